#!/bin/bash
export VS_LAB=1    # the library reads its measurement switches only then
# ab_plane.sh <outdir>: bench.py --workload configs2 with the plane kernel (1) and the general table kernel (0), alternating (same box)
OUT=$1; mkdir -p $OUT
for rep in 1 2; do for v in 1 0; do
  VS_WARP_PLANE_KERNEL=$v python3 bench.py --workload configs2 > $OUT/c2_plane${v}_$rep.json 2>> $OUT/err.log
  python3 - <<PY
import json
b=json.load(open("$OUT/c2_plane${v}_$rep.json"))["nv12_stabilize"]
print("plane_kernel=$v rep $rep: %.0f f/s warp %.1f us frac %.4f" % (b["value"], b["roofline"]["avg_launch_us"], b["roofline"]["frac"]))
PY
done; done
