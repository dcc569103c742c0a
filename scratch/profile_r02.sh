#!/bin/bash
# profile_r02.sh <tag>: the three rocprofv3 passes over `python3 bench.py` whose summaries go to profiles/ (run on the GPU box
# through gpurun; each pass is its own run: kernel trace + stats, then one PMC pass per counter, as MI355X_MICROARCH.md asks).
set -e
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_line_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 "$ROOT/bench.py" > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline > "$OUT/write.json" 2> "$OUT/write.err"
cd "$ROOT"
BT=$(python3 -c "import json,sys; print(json.load(open('$OUT/bench_line_under_rocprof.json'))['config']['build'])")
python3 profiles/summarize.py "$TAG" "$OUT/stats" "$OUT/fetch" "$OUT/write" "$BT" > "$OUT/summary.txt"
cp profiles/${TAG}_kernel_stats.csv profiles/warp_traffic.json "$OUT/"
# the raw counter csv files are large: keep the per-kernel stats and the summaries only
rm -rf "$OUT/fetch" "$OUT/write"; find "$OUT/stats" -name "*kernel_trace.csv" -delete
