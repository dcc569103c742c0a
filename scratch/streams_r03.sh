#!/bin/bash
# streams_r03.sh <outdir>: total frames/s of 1 stream, 8 streams as one vs_batch group, 8 independent instances (same box)
OUT=$1; mkdir -p $OUT
for cfg in "1 1" "8 1" "8 0" "8 1"; do set -- $cfg
  python3 bench.py --streams $1 --group $2 --no-extras --no-cpu-baseline --clip-frames 64 > $OUT/s$1g$2.json 2>> $OUT/err.log
  python3 - <<PY
import json
a=json.load(open("$OUT/s$1g$2.json"))
print("streams $1 group $2: %.0f f/s total, ms/step %.4f, frames/step/stream %d, warp %.1f us frac %.4f" % (a["value"], a["ms_per_step"], a["config"]["frames_per_step"], a["roofline"]["avg_launch_us"], a["roofline"]["frac"]))
PY
done
