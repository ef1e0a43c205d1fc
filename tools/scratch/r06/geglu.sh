# geglu_direct<320>: 80-row workgroups (0) vs 160 rows on 8 waves (1) vs 160 rows on 4 waves, fragments reused (2, SEQ): parity, isolated, in the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for v in 0 1 2; do
  FMC_GEGLU320_ROWS160=$v python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu320
done
FMC_GEGLU320_ROWS160=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_direct" 2>&1 | tail -3
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 2 0 2; do
  FMC_GEGLU320_ROWS160=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows160=$v', d['ms_per_step'])"
done
FMC_GEGLU320_ROWS160=2 timeout 900 python bench.py --no-cpu-baseline --no-fp32-line --no-loop50 2>$O/bench.err | grep '^{' > $O/bench_seq.json
python - <<PY
import json
d = json.load(open("$O/bench_seq.json"))
print("ms/step", d["ms_per_step"])
for k, v in d.items():
    if k.startswith("roofline") and v:
        print(k, v.get("kernel", "")[:50], "frac", v.get("frac"), "isolated", v.get("frac_isolated"), "in-step ms", v.get("in_step_avg_ms"), "iso ms", v.get("avg_launch_ms"), v.get("in_step_note", ""))
f = d["in_step_kernel_families"]
for k in f:
    if k.endswith("launches"): print(k, f[k])
print(f["ms_per_step"])
PY
