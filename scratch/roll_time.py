"""Time vs_roll_correct_dev on resident 4K frames: python scratch/roll_time.py [frames]"""
import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi
import roll_scene
vs = capi.load()
W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed, lines in ((1, 60), (2, 200)):
    f = roll_scene.horizon_frame(W, H, lines, seed=seed)
    d_f, d_r = capi.DevBuf.from_array(vs, f), capi.DevBuf(vs, f.nbytes)
    rc = vs.roll_correction()
    for _ in range(10):
        rc.correct_dev(d_f.ptr, W, H, W * 3, d_r.ptr, W * 3)
    rc.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        rc.correct_dev(d_f.ptr, W, H, W * 3, d_r.ptr, W * 3)
    rc.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"scene seed={seed}: {dt * 1e3:.3f} ms/frame  state={rc.state()}")
