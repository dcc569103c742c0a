#!/bin/bash
# round 4, call ar: arrival of a zoom batch by a polled word behind the masks (cur) against the event (prev)
O=gpurun_out/r04_ar; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_azc.py tests/test_gpu_pipeline.py -m gpu -x -q -k "azc or zoom or chain" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3; do for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'], d['zoom_worker_us_per_frame'], d['zoom_batch_us'])" | tee -a $O/summary.txt
done; done
for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" | tee -a $O/pairs.txt; timeout -k 10 300 python3 scratch/pair_probe.py 2>> $O/err.log | tee -a $O/pairs.txt
done
echo done
