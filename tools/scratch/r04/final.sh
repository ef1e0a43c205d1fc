cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f5; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; cp gpurun_out/roofline_counters.json profiles/roofline_counters.json; cp gpurun_out/roofline_counters.json $O/
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/trace -name "*.csv" -size +1M -delete
head -12 $O/kernel_summary.md | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
