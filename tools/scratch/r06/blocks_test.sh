cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -s -m gpu -k "every_block" 2>&1 | tail -25
