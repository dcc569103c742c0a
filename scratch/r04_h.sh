#!/bin/bash
# round 4, call h: UV lanes as two runs of two pixels (cur) against the build before (r4g), same box; roll with eight workers; chain
O=gpurun_out/r04_h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_roll.py -m gpu -x -q > $O/t1.log 2>&1; tail -3 $O/t1.log
python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "nv12 or config3" > $O/t2.log 2>&1; tail -3 $O/t2.log
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=b.get("nv12_stabilize", b); r=b["roofline"]
    print("%-28s %9.0f f/s  warp %.1f us frac %.4f" % (sys.argv[2], b["value"], r["avg_launch_us"], r["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do for n in cur r4g; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  timeout -k 10 200 python3 bench.py --workload configs2 --regions 3 > $O/c2_${n}_$rep.json 2>> $O/err.log; line $O/c2_${n}_$rep.json "c2 $n $rep" | tee -a $O/summary.txt
done; done
unset VS_LIB
python scratch/chain_probe.py 2>&1 | tee $O/chain_probe.txt
VS_BENCH_CHAIN=1 python bench.py --workload configs2 --regions 3 > $O/chain.json 2> $O/chain.err
python3 -c "
import json
d=json.loads(open('$O/chain.json').read().strip().splitlines()[-1]); print('chain', d['chain_nv12']['value'], d['chain_nv12']['ms_per_frame'])" | tee -a $O/summary.txt
echo done
