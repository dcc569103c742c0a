#!/bin/bash
# round 4, call ay: hysteresis workgroups of 16 (cur) / 8 / 4 waves: placement beside the stabilizer's launches
O=gpurun_out/r04_ay; mkdir -p $O
for n in cur hb8 hb4 acc256; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe_rs.py 2>> $O/err.log | tee -a $O/pairs.txt
done
for rep in 1 2; do for n in cur hb8 hb4 acc256; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
echo done
