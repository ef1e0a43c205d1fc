cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python tools/scratch/r04/probe_lock.py > $O/probe_lock.txt 2>&1; echo rc=$?; cat $O/probe_lock.txt | cut -c1-400
