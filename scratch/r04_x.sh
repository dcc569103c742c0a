#!/bin/bash
# round 4, call x: fewer launches per frame in the chain (one crop-and-scale launch per zoom batch, all hysteresis passes of a roll
# batch in one launch, one table launch per rotation batch) - tests, then the chain against the build before (same box)
O=gpurun_out/r04_x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_roll.py tests/test_azc.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "roll or azc or zoom or chain or nv12 or canny" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3; do for n in cur old; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_spin.so; fi
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json,sys; d=json.loads(open('$O/chain_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n $rep', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
done; done
unset VS_LIB
timeout -k 10 120 python3 scratch/chain_probe.py > $O/probe.txt 2>&1; tail -7 $O/probe.txt
echo done
