#!/bin/bash
# round 4, call u: luma / chroma tiles interleaved 2 : 1 inside a frame (mix) against the frame's luma tiles first (cur), same box
O=gpurun_out/r04_u; mkdir -p $O
VS_LIB=$PWD/scratch/labs/libvs_mix.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nv12" > $O/t_mix.log 2>&1; tail -2 $O/t_mix.log
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=b.get("nv12_stabilize", b); r=b["roofline"]
    print("%-28s %9.0f f/s  warp %.1f us frac %.4f" % (sys.argv[2], b["value"], r["avg_launch_us"], r["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do for n in cur mix; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --workload configs2 --regions 3 > $O/c2_${n}_$rep.json 2>> $O/err.log; line $O/c2_${n}_$rep.json "c2 $n $rep" | tee -a $O/summary.txt
done; done
echo done
