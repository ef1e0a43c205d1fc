cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
for v in 0 1; do
  export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_$v.json
  FMC_XATTN_FUSED_640=$v timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace$v -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace$v.log 2>&1
  T=$(find $O/trace$v -name "*kernel_trace.csv" | head -1)
  python tools/summarize_trace.py $T > $O/kernel_summary_$v.md 2>&1
  find $O/trace$v -name "*.csv" -size +1M -delete
done
