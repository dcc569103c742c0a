#!/bin/bash
# the driver's command (--steps 20 --warmup 5) over the length of the run-in in front of it
for i in 1 2 3; do for n in 200 600 1500; do
  v=$(VS_BENCH_PREROLL_BATCHES=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])")
  echo "preroll $n: $v"
done; done
