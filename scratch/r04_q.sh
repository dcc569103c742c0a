#!/bin/bash
# round 4, call q: the round's profiles - default bench line, driver-args line, rocprofv3 kernel stats and PMC passes of configs1 / configs2
O=gpurun_out/r04_q; mkdir -p $O
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_line_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_args.json 2> $O/bench_driver.err
bash scratch/profile_r03.sh r04_q all > $O/profile.log 2>&1
cat $O/progress.txt
tail -5 $O/summary.txt
echo done
