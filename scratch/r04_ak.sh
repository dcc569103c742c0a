#!/bin/bash
# round 4, call ak: the stabilizer's streams confined to all but 32 / 64 compute units (4 / 8 per XCD), so that the short kernels of
# the roll and zoom stages always find free units: pairwise probe and the chain, same box
O=gpurun_out/r04_ak; mkdir -p $O
for n in cur cum32 cum64; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
done
for rep in 1 2 3; do for n in cur cum32 cum64; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
echo done
