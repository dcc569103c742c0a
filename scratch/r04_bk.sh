#!/bin/bash
# round 4, call bk: the round's final records - full GPU suite, default bench line, driver-args line, rocprofv3 kernel stats and PMC
# passes of configs1 / configs2, all on one build
O=gpurun_out/r04_bk; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err || exit 1; tail -c 400 $O/bench_line_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_args.json 2> $O/bench_driver.err || exit 1
bash scratch/profile_r03.sh r04_bk all > $O/profile.log 2>&1
cat $O/progress.txt
tail -5 $O/summary.txt
echo done
