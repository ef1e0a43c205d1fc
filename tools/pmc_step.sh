# HBM-side traffic of one whole denoising step, by kernel (three runs of bench.py: kernel trace, FETCH_SIZE, WRITE_SIZE).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_step; mkdir -p $O
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50"
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- $B > $O/trace.log 2>&1
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- $B > $O/fetch.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- $B > $O/write.log 2>&1
python tools/summarize_pmc_step.py $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") \
    $(find $O/trace -name "*kernel_trace.csv") 2 > $O/summary.md
find $O -name "*.csv" -size +1M -delete
head -50 $O/summary.md | cut -c1-220
