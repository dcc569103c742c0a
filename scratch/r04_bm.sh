#!/bin/bash
# round 4, call bm: the full GPU suite and the default bench line on the round's last commit
O=gpurun_out/r04_bm; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc = 0 ] || { echo "tests failed"; exit 1; }
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err || exit 1
python3 -c "
import json; d=json.loads(open('$O/bench_line_default.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['configs']['configs[2]']
print(d['value'], r['frac'], r['traffic'], 'c2', c['nv12_stabilize']['value'], c['nv12_stabilize']['roofline']['frac'], c['nv12_stabilize']['roofline']['traffic'], 'chain', c['chain_nv12']['value'], 'checked', d['outputs_checked'])"
echo done
