#!/bin/bash
# round 4, call aa: hardware queues.  The chain drives ten HIP streams (roll 3 + 1, stabilizer 4, zoom 2); the runtime maps them onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run in submission order
O=gpurun_out/r04_aa; mkdir -p $O
for rep in 1 2; do for q in 4 8 12 16; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_q${q}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_q${q}_$rep.json').read().strip().splitlines()[-1]); print('queues $q:', d['value'], d['stage_thread_ms_per_chunk'], 'cores busy', d['host_cores_busy'])" | tee -a $O/summary.txt
done; done
for q in 4 8 12; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --regions 5 > $O/c1_q$q.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/c1_q$q.json').read().strip().splitlines()[-1]); print('configs1 queues $q:', d['value'], d['roofline']['frac'])" | tee -a $O/summary.txt
done
echo done
