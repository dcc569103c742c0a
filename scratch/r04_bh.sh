#!/bin/bash
# round 4, call bh: the tracker at three waves per SIMD (lk3) against four (cur): a fourth of the vector registers left to the gray kernel
O=gpurun_out/r04_bh; mkdir -p $O
export VS_LIB=$PWD/scratch/labs/libvs_lk3.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lk" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
for rep in 1 2 3 4; do for n in cur lk3; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 5 > $O/c1_${n}_$rep.json 2>> $O/err.log || exit 1
  python3 -c "import json; d=json.loads(open('$O/c1_${n}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$n $rep', d['value'], d['ms_per_step'], r['avg_launch_us'])" | tee -a $O/summary.txt
done; done
OUT=$PWD/$O; ROOT=$PWD
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
export VS_LIB=$ROOT/scratch/labs/libvs_lk3.so
rocprofv3 --kernel-trace --stats -d $OUT/tl --output-format csv -- python3 $ROOT/bench.py --no-extras --no-cpu-baseline --clip-frames 32 --regions 2 > $OUT/tl.json 2>> $OUT/err.log
python3 $ROOT/scratch/timeline.py $OUT/tl > $OUT/timeline_lk3.txt 2>&1
find $OUT/tl -name "*kernel_trace.csv" -delete
cut -c1-90 $OUT/timeline_lk3.txt
echo done
