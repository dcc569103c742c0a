#!/bin/bash
# round 4, call ao: hardware queues on the final chain, three alternating repetitions
O=gpurun_out/r04_ao; mkdir -p $O
for rep in 1 2 3 4; do for q in 4 8 12; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_q${q}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_q${q}_$rep.json').read().strip().splitlines()[-1]); print('queues $q', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
echo done
