#!/bin/bash
# round 4, call bi: the quarter-size gray kernel (cur) against the general resize kernel (prev): configs[2] (3840x2160 NV12) and the chain
O=gpurun_out/r04_bi; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_roll.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "roll or nv12 or 4k or chain or resize or gray" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3; do for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  VS_BENCH_CHAIN=1 timeout -k 10 300 python3 bench.py --workload configs2 --regions 3 > $O/c2_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "
import json; d=json.loads(open('$O/c2_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep c2', d['nv12_stabilize']['value'], d['nv12_stabilize']['roofline']['frac'], 'chain', d['chain_nv12']['value'], d['chain_nv12']['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
done
echo done
