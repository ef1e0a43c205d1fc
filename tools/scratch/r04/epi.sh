cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base v1 v2 base; do
  if [ $v = base ]; then unset FMC_HIP_LIB; else export FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_$v.so; fi
  echo "== $v"; timeout 300 python tools/scratch/probe_ffblk.py 2>&1 | grep "^M=" | cut -c1-330
done
