#!/bin/bash
export VS_LAB=1    # the library reads its measurement switches only then
# plane_skel.sh <outdir> "<kernel> <skip> [<tiles per run>]" ...: per-kernel times (rocprofv3 --stats) of the 4K NV12 stream with a plane kernel's halves switched off
OUT=$(pwd)/$1; shift; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_4K_WARM=10
for v in "$@"; do set -- $v
  VS_WARP_PLANE_RUN=${3:-2} VS_WARP_PLANE_KERNEL=$1 VS_WARP_LAB_SKIP=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/k$1s$2r${3:-2} --output-format csv -- python3 $ROOT/bench.py --workload configs2 --regions 2 > $OUT/k$1s$2r${3:-2}.json 2>> $OUT/err.log
  echo "plane kernel $1 skip $2 run ${3:-2}:"; python3 $ROOT/scratch/kavg.py $OUT/k$1s$2r${3:-2} warp_plane 2>/dev/null
done
