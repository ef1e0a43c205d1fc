cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
for v in 0 1 0 1; do
  echo "== defer=$v"; FMC_GEGLU_DEFER=$v timeout 300 python tools/scratch/probe_ffblk.py 2>&1 | grep "^M=" | cut -c1-330
done
FMC_GEGLU_DEFER=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "geglu or lnc or 160 or ffblk" > $O/pytest_defer.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_defer.log | cut -c1-300
