#!/bin/bash
# round 4, call z: the chain's copies (1 MB of mask per frame to the host), CPU accounting, PCIe rate of the box
O=gpurun_out/r04_z; mkdir -p $O
scratch/d2h_bw | tee $O/d2h_bw.txt
for rep in 1 2; do
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['stage_thread_ms_per_chunk'], 'cores busy', d['host_cores_busy'], d['cgroup_cpu'])" | tee -a $O/summary.txt
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o chain -- python3 $GRAFT_REPO_ROOT/scratch/chain_only.py 3 2 > $GRAFT_REPO_ROOT/$O/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
grep '^{' $O/trace_run.log | cut -c1-200
python3 scratch/copy_summary.py $O/trace | tee $O/copies.txt
python3 scratch/trace_overlap.py $O/trace > $O/overlap.txt 2>&1; tail -5 $O/overlap.txt
echo done
