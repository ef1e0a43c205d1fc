# round-3: remaining GPU tests after the W^T cache fix, roofline counters, the other BASELINE configs' bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_multi.py "tests/test_gpu_kernels.py::test_linear_backward_data_cache_dies_with_its_weight" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; tail -3 $O/counters.log
for c in lora cam train32; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; tail -c 600 $O/bench_$c.json | head -c 600; echo
done
timeout 600 python bench.py --mode train --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; tail -c 400 $O/bench_train.json
