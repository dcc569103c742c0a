#!/bin/bash
# same-box A/B of an environment switch: scratch/ab_env.sh VAR [rounds]   (bench lines -> gpurun_out/ab_<VAR>.txt)
VAR=$1; N=${2:-3}
out=gpurun_out/ab_$VAR.txt; : > $out
for i in $(seq $N); do
  echo "default" >> $out
  python bench.py --no-cpu-baseline --no-extras >> $out 2>&1 || exit 1
  echo "$VAR=1" >> $out
  env $VAR=1 python bench.py --no-cpu-baseline --no-extras >> $out 2>&1 || exit 1
done
python - "$out" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print(tag, d['value'], d['roofline']['frac'])
    elif line: tag = line
PY
