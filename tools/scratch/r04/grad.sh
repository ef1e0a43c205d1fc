cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -s -m gpu -k "training_gradients" 2>&1 | grep -E "gradient rel-inf|passed|failed" | tee $O/grad.log
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -s -m gpu > $O/pytest_full.log 2>&1; echo "full rc=$?"; grep -E "rel-inf|format|passed|failed|Error" $O/pytest_full.log | cut -c1-220
