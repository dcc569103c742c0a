#!/bin/bash
# kstat_variants.sh <frames> lib1.so lib2.so ...: average / minimum kernel durations of scratch/warp_one.py under rocprofv3 per library
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
F=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  D=$R/gpurun_out/kv_$(basename $L .so)
  rm -rf $D
  VS_LIB=$R/$L rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python3 $R/scratch/warp_one.py $F > /dev/null 2>&1
  python3 - "$D" "$L" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('vsd::(anonymous namespace)::', '').split('(')[0]
        print("%-24s %-34s calls %3s avg %8.1f us  min %8.1f us" % (sys.argv[2].split('/')[-1], n[:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
