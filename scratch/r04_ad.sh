#!/bin/bash
# round 4, call ad: chunk size of the free-running chain, zoom workers
O=gpurun_out/r04_ad; mkdir -p $O
run() {  # name chunk timed warm zoomworkers
  export VS_BENCH_CHAIN_CHUNK=$2 VS_AZC_WORKERS=$5
  timeout -k 10 300 python3 scratch/chain_only.py $3 $4 > $O/chain_$1.json 2>> $O/err.log || { echo "$1 failed"; tail -3 $O/err.log; return; }
  python3 -c "import json,sys; d=json.loads(open('$O/chain_$1.json').read().strip().splitlines()[-1]); print('%-16s' % '$1', d['value'], d['stage_thread_ms_per_chunk'], d['host_cores_busy'])" | tee -a $O/summary.txt
}
for rep in 1 2; do
  run c128_z8_$rep 128 8 4 8
  run c256_z8_$rep 256 4 2 8
  run c512_z8_$rep 512 3 2 8
  run c256_z12_$rep 256 4 2 12
  run c128_z12_$rep 128 8 4 12
done
echo done
