#!/bin/bash
# round 4, call bd: the warps without their wait for the detector's wide launches (nonms) against with it (cur); then configs2 + chain of cur
O=gpurun_out/r04_bd; mkdir -p $O
for rep in 1 2 3 4; do for n in cur nonms; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 5 > $O/c1_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json; d=json.loads(open('$O/c1_${n}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$n $rep', d['value'], d['ms_per_step'], r['avg_launch_us'], [round(x/1000,1) for x in d['config'].get('region_values', [])][:9])" | tee -a $O/summary.txt
done; done
unset VS_LIB
VS_BENCH_CHAIN=1 timeout -k 10 300 python3 bench.py --workload configs2 --regions 3 > $O/c2.json 2>> $O/err.log && python3 -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('c2', d['nv12_stabilize']['value'], d['nv12_stabilize']['roofline']['frac'], 'chain', d['chain_nv12']['value'], d['chain_nv12']['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
timeout -k 10 300 python3 bench.py --streams 8 --no-extras --no-cpu-baseline --regions 3 > $O/s8.json 2>> $O/err.log && python3 -c "
import json; d=json.loads(open('$O/s8.json').read().strip().splitlines()[-1]); print('streams 8 group', d['value'])" | tee -a $O/summary.txt
echo done
