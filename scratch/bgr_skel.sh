#!/bin/bash
# bgr_skel.sh <outdir>: per-kernel time of warp_tab_kernel in the headline stream with its halves switched off (VS_WARP_LAB_SKIP)
export VS_LAB=1
OUT=$(pwd)/$1; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
for v in 0 1 2 3; do
  VS_WARP_LAB_SKIP=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/s$v --output-format csv -- python3 $ROOT/bench.py --no-extras --no-cpu-baseline --regions 2 > $OUT/s$v.json 2>> $OUT/err.log
  echo "skip $v:"; python3 $ROOT/scratch/kavg.py $OUT/s$v warp_tab 2>/dev/null
done
