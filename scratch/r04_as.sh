#!/bin/bash
# round 4, call as: one batch on the GPU for one stream, eight streams as a vs_batch, eight independent instances (own clips): where
# does the group lose its 8 - 10 %?
OUT=$(pwd)/gpurun_out/r04_as; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp VS_BENCH_PREROLL_BATCHES=20
for cfg in "1 1" "8 1" "8 0"; do set -- $cfg
  rocprofv3 --kernel-trace --stats -d $OUT/s$1g$2 --output-format csv -- python3 $ROOT/bench.py --streams $1 --group $2 --no-extras --no-cpu-baseline --clip-frames 32 --regions 2 > $OUT/s$1g$2.json 2>> $OUT/err.log
  python3 $ROOT/scratch/timeline.py $OUT/s$1g$2 > $OUT/timeline_s$1g$2.txt 2>&1
  python3 $ROOT/scratch/kavg.py $OUT/s$1g$2 "" > $OUT/kavg_s$1g$2.txt 2>&1
  python3 -c "import json; d=json.loads(open('$OUT/s$1g$2.json').read().strip().splitlines()[-1]); print('streams $1 group $2:', d['value'], d['ms_per_step'])"
  find $OUT/s$1g$2 -name "*kernel_trace.csv" -size +10M -delete
done
echo done
