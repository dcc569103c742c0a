import sys, time
sys.path.insert(0, 'video-stab_amd')
import numpy as np
from vsamd import capi, synth
vs = capi.load()
W, H = 1920, 1080
fb = W * H * 3
frames = synth.make_clip(synth.SEED_CONFIG2, W, H, 12)
d_in = capi.DevBuf(vs, fb * 12)
for i, f in enumerate(frames): d_in.upload(f, i * fb)
for WB in (16, 32):
    p = vs.params(smoothing_radius=30, max_corners=200, lk_win_size=21, lk_max_level=2)
    s = vs.stabilizer(p); s.set_batch(WB); s.set_zero_copy(True)
    outs = [capi.DevBuf(vs, fb) for _ in range(96)]
    order = [i % 12 if (i // 12) % 2 == 0 else 11 - i % 12 for i in range(2000)]
    for i in range(200):
        s.push_dev(d_in.ptr + order[i] * fb, W, H, W * 3, 0, outs[i % 96].ptr, W * 3)
    s.sync()
    for n in (50, 200, 800):
        t0 = time.perf_counter()
        for i in range(200, 200 + n):
            s.push_dev(d_in.ptr + order[i] * fb, W, H, W * 3, 0, outs[i % 96].ptr, W * 3)
        t1 = time.perf_counter()
        s.sync()
        t2 = time.perf_counter()
        print("WB %d n=%d: enqueue %.1f us/frame, total %.1f us/frame" % (WB, n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    s.close()
