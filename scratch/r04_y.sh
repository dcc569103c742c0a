#!/bin/bash
# round 4, call y: the chain, same box - cur (work tree) / nofuse (hysteresis one launch per pass) / old (HEAD before the launch
# work and the contour speed-up), roll workers 3 and 6, zoom workers 8 and 12
O=gpurun_out/r04_y; mkdir -p $O
run() {   # name lib roll-workers zoom-workers
  if [ $2 = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$2.so; fi
  export VS_ROLL_WORKERS=$3 VS_AZC_WORKERS=$4
  timeout -k 10 200 python3 scratch/chain_only.py 8 4 > $O/chain_$1.json 2>> $O/err.log || { echo "$1 failed"; return; }
  python3 -c "import json,sys; d=json.loads(open('$O/chain_$1.json').read().strip().splitlines()[-1]); print('%-22s' % '$1', d['value'], d['stage_thread_ms_per_chunk'])" | tee -a $O/summary.txt
}
for rep in 1 2; do
  run old_$rep spin 3 8
  run nofuse_w3_$rep nofuse 3 8
  run cur_w3_$rep cur 3 8
  run nofuse_w6_$rep nofuse 6 8
  run cur_w6_$rep cur 6 8
  run cur_w6_z12_$rep cur 6 12
  run nofuse_w6_z12_$rep nofuse 6 12
done
for lib in spin nofuse cur; do
  if [ $lib = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$lib.so; fi
  for rw in 3 6; do
    export VS_ROLL_WORKERS=$rw VS_AZC_WORKERS=8
    echo "== $lib roll workers $rw" >> $O/probe.txt; timeout -k 10 120 python3 scratch/chain_probe.py 2>&1 | grep -v "^last" | tail -n +2 | sed -n '2,3p;5,6p' >> $O/probe.txt
  done
done
cat $O/probe.txt
echo done
