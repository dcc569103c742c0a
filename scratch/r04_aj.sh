#!/bin/bash
# round 4, call aj: zoom batch slots in flight (4 / 8 / 16), 5 roll workers, 8 / 12 zoom workers
O=gpurun_out/r04_aj; mkdir -p $O
export VS_ROLL_WORKERS=5
for rep in 1 2; do for n in cur nbs8 nbs16; do for zw in 8 12; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  export VS_AZC_WORKERS=$zw
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_z${zw}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_z${zw}_$rep.json').read().strip().splitlines()[-1]); print('$n zoom workers $zw:', d['value'], d['stage_thread_ms_per_chunk'], d['zoom_worker_us_per_frame'])" | tee -a $O/summary.txt
done; done; done
echo done
