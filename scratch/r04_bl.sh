#!/bin/bash
# round 4, call bb: (+ one upload per step, fewer event packets on `pre`) the batched warps on `pre` behind the pyramid (cur) against on `main` with events both ways (prev)
# detector's launches?  Pipeline tests, then alternating bench runs and one timeline each
O=gpurun_out/r04_bb; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3 4; do for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 5 > $O/c1_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json; d=json.loads(open('$O/c1_${n}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$n $rep', d['value'], d['ms_per_step'], r['avg_launch_us'])" | tee -a $O/summary.txt
done; done
OUT=$PWD/$O; ROOT=$PWD
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
for n in cur prev; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$ROOT/scratch/labs/libvs_$n.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/tl_$n --output-format csv -- python3 $ROOT/bench.py --no-extras --no-cpu-baseline --clip-frames 32 --regions 2 > $OUT/tl_$n.json 2>> $OUT/err.log
  python3 $ROOT/scratch/timeline.py $OUT/tl_$n > $OUT/timeline_$n.txt 2>&1
  python3 $ROOT/scratch/kavg.py $OUT/tl_$n "" 2>&1 | grep -E "tail|release|warp_tab|lk_batch|half_bgr|pyr_level" | cut -c1-40,75-140 | sed "s/^/$n /"
  find $OUT/tl_$n -name "*kernel_trace.csv" -delete
done
echo done
