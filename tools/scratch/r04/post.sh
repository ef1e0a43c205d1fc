cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "small_m or 8phase_arms or linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; cp gpurun_out/roofline_counters.json $O/; cp gpurun_out/roofline_counters.json profiles/roofline_counters.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
