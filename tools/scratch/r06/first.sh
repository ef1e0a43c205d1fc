# round 6, first GPU call: new tests, the bench line with the new objects, cold-cache autotune A/B, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention_generic or vendor_linear" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "vae or clip or pipeline_end" >> $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1200 python bench.py --autotune-log $O/autotune.log > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "steps/s", d["value"], "parity", d["parity_rel_inf"])
for k, v in d.items():
    if k.startswith("roofline") and v:
        print(k, v.get("kernel", "")[:50], "frac", v.get("frac"), "isolated", v.get("frac_isolated"), "in-step ms", v.get("in_step_avg_ms"), "iso ms", v.get("avg_launch_ms"), v.get("in_step_note", ""))
print(d.get("in_step_source")); print(json.dumps(d.get("in_step_kernel_families"))); print(json.dumps(d.get("step_dispatch")))
print("fp32", json.dumps(d.get("fp32_parity_mode"))); print("loop50", d.get("ddim_50_step_loop_s"), d.get("ddim_50_step_loop_steps_per_s"))
PY
# A/B on this box: (a) fresh WARM tune without the tracked table, (b) fresh COLD tune
for v in warm cold; do
  export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_$v.json
  if [ $v = cold ]; then export FMC_TUNE_COLD=1; fi
  FMC_AUTOTUNE_DEFAULTS=0 timeout 1500 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 --autotune-log $O/autotune_$v.log 2>$O/err_$v.log | grep '^{' > $O/bench_$v.json
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['ms_per_step'], d['autotune'])"
done
unset FMC_TUNE_COLD
# the cold table on the default (warm) path, second run: which table wins when only the table differs
FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cold.json FMC_AUTOTUNE_DEFAULTS=0 timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cold table, 2nd run', d['ms_per_step'], d['autotune'])"
FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_warm.json FMC_AUTOTUNE_DEFAULTS=0 timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warm table, 2nd run', d['ms_per_step'], d['autotune'])"
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 3000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
