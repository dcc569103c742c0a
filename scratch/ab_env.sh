#!/bin/bash
# ab_env.sh <outdir> <VAR> : bench.py --no-extras with and without a lab switch, alternating (same box)
export VS_LAB=1
OUT=$1; V=$2; mkdir -p $OUT
for rep in 1 2 3; do for on in 0 1; do
  if [ $on = 1 ]; then export $V=1; else unset $V; fi
  python3 bench.py --no-extras --no-cpu-baseline > $OUT/${V}_${on}_$rep.json 2>> $OUT/err.log
  python3 -c "
import json; d=json.load(open('$OUT/${V}_${on}_$rep.json')); print('$V=%d rep $rep: %.0f f/s  %.4f ms/step  warp %.1f us frac %.4f' % ($on, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; done
