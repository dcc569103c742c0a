#!/bin/bash
# round 4, call av: BGR warp, tap address as shift + mask + one multiply-add (cur) against multiply + shift-add (prev), same box
O=gpurun_out/r04_av; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "warp" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3 4; do for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 5 > $O/c1_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json; d=json.loads(open('$O/c1_${n}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$n $rep', d['value'], r['avg_launch_us'], r['frac'])" | tee -a $O/summary.txt
done; done
echo done
