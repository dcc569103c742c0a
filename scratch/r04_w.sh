#!/bin/bash
# round 4, call w: the chain's host side - cores of the box, spinning against blocking waits (same box), HIP API trace of the chain
O=gpurun_out/r04_w; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep Cpus_allowed_list /proc/self/status; } > $O/cpus.txt 2>&1; cat $O/cpus.txt
for rep in 1 2 3; do for n in cur spin; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'], d['whole_run'])" | tee -a $O/summary.txt
done; done
unset VS_LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o chain -- python3 $GRAFT_REPO_ROOT/scratch/chain_only.py 3 2 > $GRAFT_REPO_ROOT/$O/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scratch/hip_api_summary.py $O/trace > $O/hip_api_summary.txt 2>&1; head -50 $O/hip_api_summary.txt
python3 scratch/trace_overlap.py $O/trace > $O/overlap.txt 2>&1; tail -5 $O/overlap.txt
find $O/trace -name "*_trace.csv" -size +8M -delete
echo done
