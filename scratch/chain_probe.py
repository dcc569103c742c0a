"""Where the configs[2] chain spends its time: each asynchronous stage alone on 4K NV12 surfaces (host enqueue time and total)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-stab_amd"))
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
W, H, NF, N = 3840, 2160, 32, 256
sb = W * H * 3 // 2
clip = synth.make_clip_dev(vs, synth.SEED_CONFIG3, W, H, NF, nv12=True)
d_out = capi.DevBuf(vs, sb * 64)
rc, az = vs.roll_correction(), vs.auto_zoom_crop()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(N):
        rc.correct_nv12_dev(clip.ptr + (i % NF) * sb, W, H, W, d_out.ptr + (i % 64) * sb, W)
    t1 = time.perf_counter()
    rc.sync()
    t2 = time.perf_counter()
    print("roll : enqueue %.1f us/frame, total %.1f us/frame (%.0f f/s)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, N / (t2 - t0)))
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(N):
        az.apply_nv12_dev(clip.ptr + (i % NF) * sb, W, H, W, d_out.ptr + (i % 64) * sb, W, W * H)
    t1 = time.perf_counter()
    az.sync()
    t2 = time.perf_counter()
    print("zoom : enqueue %.1f us/frame, total %.1f us/frame (%.0f f/s)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, N / (t2 - t0)))
print("last zoom result", az.result(N * 3 - 1)[:2])
wt = az.worker_times()
print("zoom workers, us per frame: idle %.1f masks %.1f contour %.1f publish %.1f" % tuple(v / wt[0] * 1e6 for v in wt[1:5]), "; per batch us: launch->masks %.1f masks->crop %.1f caller waits for slot %.1f" % tuple(v / wt[5] * 1e6 for v in wt[6:9]))
