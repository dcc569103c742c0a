#!/bin/bash
# ab_run.sh <outdir> "<kernel>:<run>" ...: parity tests of the plane kernels, then bench.py --workload configs2, per setting (same box)
export VS_LAB=1
OUT=$1; shift; mkdir -p $OUT
for kr in "$@"; do
  k=${kr%%:*}; r=${kr##*:}
  export VS_WARP_PLANE_KERNEL=$k VS_WARP_PLANE_RUN=$r
  if [ "$k" != "1" ]; then
    timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "plane or nv12" > $OUT/t_${k}_$r.log 2>&1 || { echo "parity FAILED for $kr"; tail -15 $OUT/t_${k}_$r.log; exit 1; }
    tail -1 $OUT/t_${k}_$r.log
  fi
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --workload configs2 --no-cpu-baseline > $OUT/c2_${k}_${r}_$rep.json 2>> $OUT/err.log || { echo "bench failed"; tail -5 $OUT/err.log; exit 1; }
    python3 - <<PY
import json
b=json.load(open("$OUT/c2_${k}_${r}_$rep.json"))
b=b.get("nv12_stabilize", b)
print("kernel=$k run=$r rep $rep: %.0f f/s warp %.1f us frac %.4f" % (b["value"], b["roofline"]["avg_launch_us"], b["roofline"]["frac"]))
PY
  done
done
