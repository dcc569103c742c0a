#!/bin/bash
# tl_now.sh <outdir>: kernel trace of the headline stream -> per-kernel averages and one steady-state batch (timeline.py)
OUT=$(pwd)/$1; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/tr --output-format csv -- python3 $ROOT/bench.py --no-extras --no-cpu-baseline --regions 2 > $OUT/line.json 2> $OUT/err.log
python3 $ROOT/scratch/kavg.py $OUT/tr "" | sort -t'g' -k2 | head -30
python3 $ROOT/scratch/timeline.py $OUT/tr 2>/dev/null | head -40
