#!/bin/bash
# round 4, call c: CU-mask probe; same-box A/B of the tables fold + NV12 merge (cur) against the commit before (r4a); CU-mask variants
O=gpurun_out/r04_c; mkdir -p $O
scratch/cu_mask_probe.bin > $O/probe.txt 2>&1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=b.get("nv12_stabilize", b); r=b["roofline"]
    print("%-28s %9.0f f/s  warp %.1f us frac %.4f stage %s" % (sys.argv[2], b["value"], r["avg_launch_us"], r["frac"], r.get("stage_frac")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do for n in cur r4a; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 3 > $O/c1_${n}_$rep.json 2>> $O/err.log; line $O/c1_${n}_$rep.json "c1 $n $rep" | tee -a $O/summary.txt
  timeout -k 10 200 python3 bench.py --workload configs2 --regions 3 > $O/c2_${n}_$rep.json 2>> $O/err.log; line $O/c2_${n}_$rep.json "c2 $n $rep" | tee -a $O/summary.txt
done; done
export VS_LIB=$PWD/scratch/labs/libvs_cum.so
run() { name=$1; shift; for rep in 1 2; do env "$@" timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --regions 3 > $O/cum_${name}_$rep.json 2>> $O/err.log; line $O/cum_${name}_$rep.json "cum $name $rep" | tee -a $O/summary.txt; done; }
run none VS_X=1
run pd8 VS_CUM_PRE=0:8 VS_CUM_DET=0:8
run pd32 VS_CUM_PRE=0:32 VS_CUM_DET=0:32
run lk32 VS_CUM_LK=0:32
run lk64 VS_CUM_LK=0:64
run lk32x VS_CUM_LK=0:32 VS_CUM_PRE=32:256 VS_CUM_DET=32:256
run lk64x VS_CUM_LK=0:64 VS_CUM_PRE=64:256 VS_CUM_DET=64:256
run lk128x VS_CUM_LK=0:128 VS_CUM_PRE=128:256 VS_CUM_DET=128:256
run lkonly VS_CUM_LK=0:0
echo done
