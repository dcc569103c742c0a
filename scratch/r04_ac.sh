#!/bin/bash
# round 4, call ac: kernel + copy trace of the free-running chain
O=gpurun_out/r04_ac; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o chain -- python3 $GRAFT_REPO_ROOT/scratch/chain_only.py 4 3 > $GRAFT_REPO_ROOT/$O/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
grep '^{' $O/trace_run.log | cut -c1-300
ls -la $O/trace
echo done
