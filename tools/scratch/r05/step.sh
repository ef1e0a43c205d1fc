# bench + kernel-trace summary of the step; usage: bash tools/scratch/r05/step.sh <tag> [extra pytest selection]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 600 python -m pytest tests/test_gpu_conv_halo.py -x -q -m gpu > $O/pytest_halo.log 2>&1; tail -3 $O/pytest_halo.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err; head -c 400 $O/bench.json; echo
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/trace -name "*.csv" -size +1M -delete
head -60 $O/kernel_summary.md | cut -c1-170
