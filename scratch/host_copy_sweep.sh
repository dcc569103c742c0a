#!/bin/bash
# pageable vs_stab_push with and without the helper thread (scratch/host_copy_rate.py) -> gpurun_out/host_copy_sweep.txt
out=gpurun_out/host_copy_sweep.txt; : > $out
export VS_RATE_ONLY_PUSH=1
for i in 1 2; do
  VS_STAB_HOST_HELPER=0 python scratch/host_copy_rate.py >> $out 2>&1 || exit 1
  python scratch/host_copy_rate.py >> $out 2>&1 || exit 1
done
grep -v "^host cores" $out
