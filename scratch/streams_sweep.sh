#!/bin/bash
# bench.py --streams S (S instances on one GPU, shared stream pool), steps of 64 frames per stream
for s in 1 2 4 8; do
  python bench.py --streams $s --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $s, d['value'], d['roofline']['frac'])"
done
