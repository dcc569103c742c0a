#!/bin/bash
# round 4, call au: the hysteresis of a roll batch by one workgroup per frame in one launch (cur) against twelve launches (prev)
O=gpurun_out/r04_au; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_roll.py tests/test_gpu_pipeline.py -m gpu -x -q -k "roll or chain or canny" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3; do for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'], d['zoom_batch_us'])" | tee -a $O/summary.txt
done; done
for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
done
echo done
