#!/bin/bash
# round 4, call t: first-chunk size of the corner selection (448 / 768 / 1400) and the pre-chunk build (r4d): select kernel time under
# rocprofv3 --stats, one stream and eight streams with clips of their own; candidate counts of the bench clips; checkerboard case
O=$PWD/gpurun_out/r04_t; mkdir -p $O
R=$PWD
python3 - > $O/cand.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "video-stab_amd")
from vsamd import capi, synth
vs = capi.load()
W, H, NF = 1920, 1080, 24
fb = W * H * 3
for g in (0, 1, 3, 5):
    clip = synth.make_clip_dev(vs, synth.SEED_CONFIG2 + g, W, H, NF)
    s = vs.stabilizer(vs.params(max_corners=200, lk_win_size=21, smoothing_radius=30))
    out = capi.DevBuf(vs, fb)
    c = []
    for i in range(NF):
        s.push_dev(clip.ptr + i * fb, W, H, W * 3, capi.FMT_BGR8, out.ptr, W * 3)
        if i % 2 == 1:
            c.append(s.counters().last_candidates)
    print("stream", g, "candidates per detection:", c)
    s.close(); clip.free(); out.free()
PY
cat $O/cand.txt
cd /tmp && export TMPDIR=/tmp
for n in r4d sel448 sel768 sel1400; do for S in 1 8; do
  VS_LIB=$R/scratch/labs/libvs_$n.so VS_BENCH_PREROLL_BATCHES=40 rocprofv3 --kernel-trace --stats -d $O/p_${n}_$S --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --regions 3 --streams $S > $O/b_${n}_$S.json 2> $O/b_${n}_$S.err
  f=$(ls $O/p_${n}_$S/*/*kernel_stats.csv | head -1)
  python3 - $f $O/b_${n}_$S.json "$n streams=$S" <<'PY'
import csv, json, sys
k = {r["Name"].split("(")[0].split("::")[-1]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-18s %8.0f f/s  select %.1f us  nms %.1f us  min_eigen %.1f us" % (sys.argv[3], b["value"], k.get("select_batch_kernel", 0), k.get("nms_batch_kernel", 0), k.get("min_eigen_batch_kernel<3>", k.get("void vsd", 0))))
PY
  find $O/p_${n}_$S -name "*kernel_trace.csv" -delete
done; done | tee $O/summary.txt
cd $R
tests/cpp/_build/wrapper_time 0 checker | tee $O/checker_cur.txt
echo done
