# full GPU validation: whole -m gpu suite, bench (with in-step trace), kernel-trace summaries (by kernel and by grid), smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python bench.py --autotune-log $O/autotune.log > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "steps/s", d["value"], "parity", d["parity_rel_inf"])
for k, v in d.items():
    if k.startswith("roofline") and v:
        print(k, v.get("kernel", "")[:50], "frac", v.get("frac"), "isolated", v.get("frac_isolated"), "in-step ms", v.get("in_step_avg_ms"), "iso ms", v.get("avg_launch_ms"))
print(d.get("in_step_source")); print(json.dumps(d.get("in_step_kernel_families"))); print(json.dumps(d.get("step_dispatch")))
PY
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --trace-child > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
python tools/summarize_trace.py $T --by-grid > $O/kernel_by_grid.md 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/trace -name "*.csv" -size +1M -delete
head -24 $O/kernel_summary.md | cut -c1-170
if [ "$2" != "notests" ]; then
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
fi
