#!/bin/bash
# round 4, call ab: a stream per zoom batch slot (cur) against one stream for all mask kernels and copies (oneq), free-running chain
O=gpurun_out/r04_ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_azc.py tests/test_roll.py tests/test_gpu_pipeline.py -m gpu -x -q -k "azc or zoom or chain or roll" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3; do for n in cur oneq; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'], d['host_cores_busy'])" | tee -a $O/summary.txt
done; done
for n in cur oneq; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  echo "== $n" >> $O/probe.txt; timeout -k 10 120 python3 scratch/chain_probe.py 2>&1 | grep -v "^last" >> $O/probe.txt
done
cat $O/probe.txt
echo done
