#!/bin/bash
# round 4, call al: pairwise probe and chain with more hardware queues (GPU_MAX_HW_QUEUES 4 = default / 12 / 24)
O=gpurun_out/r04_al; mkdir -p $O
for q in 4 12 24; do
  export GPU_MAX_HW_QUEUES=$q
  echo "== queues $q" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
done
for rep in 1 2; do for q in 4 12 24; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_q${q}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_q${q}_$rep.json').read().strip().splitlines()[-1]); print('queues $q', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
echo done
