cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "temporal_block_fused" > $O/pytest_tb.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_tb.log | cut -c1-300
timeout 120 python tools/scratch/r04/probe_tb.py > $O/probe_tb.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/probe_tb.txt | cut -c1-330
