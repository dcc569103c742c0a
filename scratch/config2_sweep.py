"""configs[2] NV12 stabilisation over batch size / warm-up / timed batches (bench.config2 with its environment knobs)."""
import os, sys, json, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'video-stab_amd'));"
        "import bench, argparse; from vsamd import capi; vs = capi.load();"
        "o = bench.config2(vs, 0, None)['nv12_stabilize']; print(o['value'], o['roofline']['frac'], o['roofline']['avg_launch_us'])") % (os.path.dirname(here), os.path.dirname(here))
for bt, warm, timed in [(16, 2, 8), (16, 40, 40), (32, 20, 20), (32, 40, 40)]:
    env = dict(os.environ, VS_BENCH_4K_BATCH=str(bt), VS_BENCH_4K_WARM=str(warm), VS_BENCH_4K_TIMED=str(timed))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("batch %2d warm %2d timed %2d:" % (bt, warm, timed), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
