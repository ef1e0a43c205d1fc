cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "temporal_block_fused_640" > $O/pytest_tb640.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_tb640.log | cut -c1-400
timeout 120 python tools/scratch/r04/probe_tb640.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
