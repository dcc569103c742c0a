#!/bin/bash
# round 4, call aq: where a zoom batch's time goes (alone and in the chain)
O=gpurun_out/r04_aq; mkdir -p $O
for rep in 1 2; do
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['stage_thread_ms_per_chunk'], d['zoom_worker_us_per_frame'], d['zoom_batch_us'])" | tee -a $O/summary.txt
done
timeout -k 10 120 python3 scratch/chain_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
echo done
