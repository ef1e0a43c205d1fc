# round 6 evidence: fresh arm table from bench runs only -> headline bench line on it -> kernel trace -> config lines -> whole-step traffic -> roofline counters
# -> full GPU suite + smoke -> cpu_baseline by SURVEY 8d's protocol.     usage: final.sh <out dir name> [notable]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
if [ "$2" != "notable" ]; then
# ---- (1) the tracked arm table, rebuilt from bench runs only (no test-suite shapes), hipBLASLt version recorded
for r in a b; do
  FMC_AUTOTUNE_DEFAULTS=0 FMC_AUTOTUNE_CACHE=$PWD/$O/tune_obj_$r.json timeout 1200 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line 2>$O/err_tune_obj_$r.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune obj $r', d['ms_per_step'], d['autotune']['shapes_tuned_in_this_run'])"
done
for c in lora cam; do
  FMC_AUTOTUNE_DEFAULTS=0 FMC_AUTOTUNE_CACHE=$PWD/$O/tune_$c.json timeout 1200 python bench.py --config $c --no-cpu-baseline --no-in-step --no-fp32-line 2>$O/err_tune_$c.log | grep '^{' > $O/line_$c.json
done
FMC_AUTOTUNE_DEFAULTS=0 FMC_AUTOTUNE_CACHE=$PWD/$O/tune_train.json timeout 1200 python bench.py --mode train --no-cpu-baseline 2>$O/err_tune_train.log | grep '^{' > $O/line_train.json
FMC_AUTOTUNE_DEFAULTS=0 FMC_AUTOTUNE_CACHE=$PWD/$O/tune_train32.json timeout 1500 python bench.py --config train32 --fp8-temporal --no-cpu-baseline 2>$O/err_tune_train32.log | grep '^{' > $O/line_train32.json
python tools/make_default_arm_table.py $O/tune_obj_a.json $O/tune_obj_b.json $O/tune_lora.json $O/tune_cam.json $O/tune_train.json $O/tune_train32.json
cp synfmc_amd/autotune_default_mi355x.json $O/autotune_default_mi355x.json
cat $O/line_lora.json $O/line_cam.json $O/line_train.json $O/line_train32.json > $O/config_bench_lines.jsonl
python -c "
import json
for l in open('$O/config_bench_lines.jsonl'):
    d=json.loads(l); print(d['config'].get('baseline_config'), d['config'].get('workload','')[:40], d['value'], d['ms_per_step'], d.get('ddim_50_step_loop_steps_per_s'))"
fi
# ---- (2) the headline line on the tracked table (fresh per-build cache), with the in-step trace, fp32 sub-line, 50-step loop
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1500 python bench.py --autotune-log $O/autotune.log > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "steps/s", d["value"], "parity", d["parity_rel_inf"], "autotune", d["autotune"])
for k, v in d.items():
    if k.startswith("roofline") and v:
        print(k, v.get("kernel", "")[:50], "frac", v.get("frac"), "isolated", v.get("frac_isolated"), "in-step ms", v.get("in_step_avg_ms"), "iso ms", v.get("avg_launch_ms"), v.get("in_step_note", ""))
f = d["in_step_kernel_families"]
for k in f:
    if k.endswith("launches"): print(k, f[k])
print(f["ms_per_step"]); print(json.dumps(d.get("step_dispatch")))
print("fp32", json.dumps(d.get("fp32_parity_mode"))); print("loop50", d.get("ddim_50_step_loop_s"), d.get("ddim_50_step_loop_steps_per_s"))
PY
# ---- (3) kernel trace of the same command (steady-state window)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --trace-child > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
python tools/summarize_trace.py $T --by-grid > $O/kernel_by_grid.md 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/trace -name "*.csv" -size +1M -delete
head -24 $O/kernel_summary.md | cut -c1-170
# ---- (4) whole-step HBM traffic by kernel
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step/summary.md $O/step_hbm_traffic_by_kernel.md; head -12 $O/step_hbm_traffic_by_kernel.md | cut -c1-200
# ---- (5) roofline counters on these kernel sources
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; cp gpurun_out/roofline_counters.json $O/; python -c "
import json; d=json.load(open('$O/roofline_counters.json'))
for k,v in d['kernels'].items(): print(k, v.get('traffic_bytes'), v.get('matrix_pipe_busy'), v.get('avg_launch_us_under_counters'), v.get('kernel_name','')[:40])"
# ---- (6) the whole GPU suite + smoke
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_tests.json
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
# ---- (7) cpu_baseline by SURVEY 8d's protocol (1 warm-up + 3 timed oracle steps)
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1500 python bench.py --cpu-baseline-full --no-in-step --no-fp32-line --no-loop50 2>$O/err_cpu_full.log | grep '^{' > $O/bench_cpu_baseline_full.json
python -c "
import json; d=json.load(open('$O/bench_cpu_baseline_full.json')); print(d['ms_per_step'], json.dumps(d['cpu_baseline'])[:600])"
