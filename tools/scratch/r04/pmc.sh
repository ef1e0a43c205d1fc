cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
export FMC_AUTOTUNE_CACHE=$PWD/gpurun_out/r04h/autotune_cache.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04h/warm.json 2>&1
bash tools/pmc_step.sh > gpurun_out/r04h/pmc_step.log 2>&1
cp gpurun_out/pmc_step/summary.md gpurun_out/r04h/step_hbm_traffic_by_kernel.md
timeout 1500 python tools/collect_roofline_counters.py > gpurun_out/r04h/counters.log 2>&1
tail -40 gpurun_out/r04h/counters.log
head -40 gpurun_out/r04h/step_hbm_traffic_by_kernel.md | cut -c1-200
