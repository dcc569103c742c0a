#!/bin/bash
# pmc_plane.sh <outdir> "<kernel> <run>" ...: dynamic instruction counts per wave of the plane kernels (own PMC pass per setting)
export VS_LAB=1
OUT=$(pwd)/$1; shift; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_4K_WARM=10 VS_BENCH_4K_TIMED=6
for v in "$@"; do set -- $v
  d=$OUT/k$1r${2:-2}; rm -rf $d
  VS_WARP_PLANE_RUN=${2:-2} VS_WARP_PLANE_KERNEL=$1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $d --output-format csv -- python3 $ROOT/bench.py --workload configs2 --regions 1 > $d.json 2>> $OUT/err.log
  echo "plane kernel $1 run ${2:-2}:"; python3 $ROOT/scratch/pmc_per_wave.py $d warp_plane
  find $d -name "*counter_collection.csv" -delete; find $d -name "*kernel_trace.csv" -delete
done
