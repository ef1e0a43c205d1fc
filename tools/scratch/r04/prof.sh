cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
find $O/trace -name "*.csv" -size +1M -delete
head -60 $O/kernel_summary.md | cut -c1-150
