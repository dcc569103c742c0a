#!/bin/bash
# round 4, call ai: worker counts on the free-running chain
O=gpurun_out/r04_ai; mkdir -p $O
for rep in 1 2; do for rw in 3 5 8; do for zw in 8 12; do
  export VS_ROLL_WORKERS=$rw VS_AZC_WORKERS=$zw
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_r${rw}_z${zw}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_r${rw}_z${zw}_$rep.json').read().strip().splitlines()[-1]); print('roll $rw zoom $zw:', d['value'], d['stage_thread_ms_per_chunk'], d['host_cores_busy'])" | tee -a $O/summary.txt
done; done; done
echo done
