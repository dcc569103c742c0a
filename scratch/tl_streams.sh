#!/bin/bash
OUT=$(pwd)/$1; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
for cfg in "1 1"; do set -- $cfg
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/s$1 --output-format csv -- python3 $ROOT/bench.py --streams $1 --group $2 --no-extras --no-cpu-baseline --clip-frames 32 --regions 2 > $OUT/s$1.json 2>> $OUT/err.log
  python3 $ROOT/scratch/timeline.py $OUT/s$1 > $OUT/timeline_s$1.txt 2>&1
  python3 $ROOT/scratch/kavg.py $OUT/s$1 "" > $OUT/kavg_s$1.txt 2>&1
done
