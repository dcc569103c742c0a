#!/bin/bash
# round 4, call e: full GPU suite on the chunked corner selection; configs[2] chain; streams 8 (own clips) with stage times; same-box A/B against r4d (before the selection change)
O=gpurun_out/r04_e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
VS_BENCH_CHAIN=1 python bench.py --workload configs2 --regions 3 > $O/chain.json 2> $O/chain.err; tail -c 700 $O/chain.json; echo
for rep in 1 2; do for n in cur r4d; do
  if [ $n = cur ]; then unset VS_LIB; else export VS_LIB=$PWD/scratch/labs/libvs_$n.so; fi
  python bench.py --no-extras --no-cpu-baseline --regions 3 --streams 8 --profile-stages > $O/s8_${n}_$rep.json 2>> $O/err.log
  python bench.py --no-extras --no-cpu-baseline --regions 3 --profile-stages > $O/s1_${n}_$rep.json 2>> $O/err.log
  python3 - $O/s8_${n}_$rep.json $O/s1_${n}_$rep.json "$n $rep" <<'PY'
import json,sys
for f,t in ((sys.argv[1],"streams8"),(sys.argv[2],"streams1")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); s=b["stage_us_per_launch"]
        print("%-8s %-10s %9.0f f/s  gftt %.1f lk %.1f traj %.1f gray %.1f pyr %.1f warp %.1f" % (sys.argv[3], t, b["value"], s["gftt"], s["lk"], s["traj"], s["gray"], s["pyramid"], s["warp"]))
    except Exception as e: print(sys.argv[3], t, "FAILED", e)
PY
done; done | tee $O/summary.txt
unset VS_LIB
tests/cpp/_build/wrapper_time 0 > $O/wrapper_time.txt 2>&1; tail -3 $O/wrapper_time.txt
echo done
