cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g256
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "256x320 or 160x320_persistent" > gpurun_out/g256/pytest.log 2>&1; tail -15 gpurun_out/g256/pytest.log
timeout 300 python tools/scratch/probe_g256.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g256/probe.txt
