import sys, time
sys.path.insert(0, 'video-stab_amd')
import numpy as np
from vsamd import capi, synth
vs = capi.load()
W, H = 1920, 1080
frames = synth.make_clip(synth.SEED_CONFIG2, W, H, 12)
p = vs.params(smoothing_radius=30, max_corners=200, lk_win_size=21, lk_max_level=2)
s = vs.stabilizer(p)
for i in range(60):
    s.push(frames[i % 12])
t0 = time.perf_counter()
n = 200
for i in range(n):
    s.push(frames[i % 12])
dt = time.perf_counter() - t0
print("host-pointer API (pageable numpy in/out, synchronous): %.1f frames/s (%.2f ms/frame)" % (n / dt, dt / n * 1e3))
