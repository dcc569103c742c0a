#!/bin/bash
# ab_c1.sh <outdir> [reps]: warp parity tests, then the headline bench line (no extras) a few times
OUT=$1; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "warp" > $OUT/t.log 2>&1 || { echo "parity FAILED"; tail -15 $OUT/t.log; exit 1; }
tail -1 $OUT/t.log
for rep in $(seq 1 ${2:-2}); do
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > $OUT/c1_$rep.json 2>> $OUT/err.log || { echo bench failed; tail -5 $OUT/err.log; exit 1; }
  python3 - <<PY
import json
b=json.load(open("$OUT/c1_$rep.json"))
r=b["roofline"]
print("rep $rep: %.0f f/s warp %.1f us frac %.4f stage_frac %s" % (b["value"], r["avg_launch_us"], r["frac"], r.get("stage_frac")))
PY
done
