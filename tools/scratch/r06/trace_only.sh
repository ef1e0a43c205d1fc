# kernel trace of the step (steady-state window), summaries only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --trace-child > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
python tools/summarize_trace.py $T --by-grid > $O/kernel_by_grid.md 2>&1
find $O/trace -name "*.csv" -size +1M -delete
head -24 $O/kernel_summary.md | cut -c1-170
grep -n "gn_fold\|gemm160p\|gn_apply\|gemm_k320" $O/kernel_by_grid.md | cut -c1-140
