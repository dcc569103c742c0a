#!/bin/bash
# round 4, call ae: where the zoom workers' time goes, alone and in the chain, 8 and 12 workers
O=gpurun_out/r04_ae; mkdir -p $O
for z in 8 12; do
  export VS_AZC_WORKERS=$z
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_z$z.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_z$z.json').read().strip().splitlines()[-1]); print('chain z$z', d['value'], d['stage_thread_ms_per_chunk'], d['zoom_worker_us_per_frame'])" | tee -a $O/summary.txt
  timeout -k 10 120 python3 scratch/chain_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
done
echo done
