#!/bin/bash
# round 4, call ah: priorities - zoom streams high, roll streams normal (cur) / both high (bothhi) / both normal (noprio)
O=gpurun_out/r04_ah; mkdir -p $O
echo "== cur" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
for rep in 1 2 3; do for n in cur bothhi noprio; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'], d['zoom_worker_us_per_frame'])" | tee -a $O/summary.txt
done; done
echo done
