#!/bin/bash
# round 4, call bj: one batch on the GPU, 3840x2160 NV12, with the quarter-size gray kernel
OUT=$PWD/gpurun_out/r04_bj; mkdir -p $OUT; ROOT=$PWD
cd /tmp; export TMPDIR=/tmp VS_BENCH_4K_WARM=20
rocprofv3 --kernel-trace --stats -d $OUT/tl --output-format csv -- python3 $ROOT/bench.py --workload configs2 --regions 2 > $OUT/tl.json 2>> $OUT/err.log
python3 $ROOT/scratch/timeline.py $OUT/tl > $OUT/timeline_c2.txt 2>&1
python3 $ROOT/scratch/kavg.py $OUT/tl "" 2>&1 | head -12 | cut -c1-40,75-140
find $OUT/tl -name "*kernel_trace.csv" -delete
cut -c1-90 $OUT/timeline_c2.txt
echo done
