#!/bin/bash
# profile_r02_all.sh <tag>: everything profiles/ keeps of a build - profile_r02.sh plus the driver's command, the stage-event
# line and the timeline of one batch.  Outputs under gpurun_out/<tag>/.
set -e
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
bash scratch/profile_r02.sh "$TAG"
echo "[profile] traffic + stats done"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line_driver_args.json" 2> "$OUT/driver.err"
python3 bench.py --no-cpu-baseline --no-extras --profile-stages > "$OUT/bench_line_with_stage_events.json" 2> "$OUT/stages.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/tl" --output-format csv -- python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline > "$OUT/tl.json" 2> "$OUT/tl.err"
cd "$ROOT"
python3 scratch/timeline.py "$OUT/tl" > "$OUT/timeline_one_batch.txt"
rm -rf "$OUT/tl"
head -3 "$OUT/timeline_one_batch.txt"
cat "$OUT/summary.txt" | tail -5
