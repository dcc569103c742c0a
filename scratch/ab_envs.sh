#!/bin/bash
# ab_envs.sh <outdir> <VAR> <value> [<value> ...]: bench.py --no-extras without VAR and with each value, three rounds, on one box
export VS_LAB=1
OUT=$1; V=$2; shift 2; mkdir -p $OUT
for rep in 1 2 3; do for x in - "$@"; do
  if [ "$x" = - ]; then unset $V; else export $V=$x; fi
  python3 bench.py --no-extras --no-cpu-baseline > $OUT/${V}_${x}_$rep.json 2>> $OUT/err.log
  python3 -c "
import json; d=json.load(open('$OUT/${V}_${x}_$rep.json')); print('$V=$x rep $rep: %.0f f/s  %.4f ms/step  warp %.1f us frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; done
