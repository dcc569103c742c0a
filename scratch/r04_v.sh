#!/bin/bash
# round 4, call v: full GPU suite on the final build, then the configs[2] chain with the array entry points
O=gpurun_out/r04_v; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
VS_BENCH_CHAIN=1 timeout -k 10 300 python bench.py --workload configs2 --regions 3 > $O/chain.json 2> $O/chain.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_v/chain.json").read().strip().splitlines()[-1])
print("c2", d["value"], d["roofline"]["frac"], "chain", d["config"].get("chain_nv12"))
PY
echo done
