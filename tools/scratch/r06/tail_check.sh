# model-level parity + the default bench line with ff_tail on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_model.py tests/test_gpu_full_width.py -q -m gpu -x 2>&1 | tail -5
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "steps/s", d["value"], "parity", d["parity_rel_inf"], "autotune", d["autotune"])
for k in ("roofline_ff2", "roofline_linear_l0", "roofline_proj_l0"):
    v = d[k]; print(k, v.get("kernel", "")[:60], "frac", v.get("frac"), "isolated", v.get("frac_isolated"), "in-step ms", v.get("in_step_avg_ms"), "iso ms", v.get("avg_launch_ms"), v.get("in_step_note", ""))
f = d["in_step_kernel_families"]
for k in f:
    if k.endswith("launches"): print(k, f[k])
print(json.dumps(d.get("step_dispatch")))
print("fp32", json.dumps(d.get("fp32_parity_mode"))[:300]); print("loop50", d.get("ddim_50_step_loop_s"), d.get("ddim_50_step_loop_steps_per_s"))
PY
