cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for st in 0 5 6; do echo "== stop $st"; FMC_TB_STOP=$st timeout 120 python tools/scratch/r04/dbg_tb.py 10 2>&1 | grep -v amdgpu.ids | grep -E "merge|per pixel|per 40"; done
