"""BASELINE configs[2]: 1 stream 3840x2160 NV12, 400 corners (per-frame pipeline), and RollCorrection + AutoZoomCrop
on 4K BGR frames (the reference applies them to BGR cv::Mat)."""
import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi, synth
import roll_scene
vs = capi.load()
W, H = 3840, 2160
clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3, W, H, 6)]
fb = clip[0].nbytes
d_in = capi.DevBuf(vs, fb * 6)
for i, f in enumerate(clip): d_in.upload(f, i * fb)
outs = [capi.DevBuf(vs, fb) for _ in range(4)]
p = vs.params(smoothing_radius=30, max_corners=400, lk_win_size=21, lk_max_level=2)
order = [i % 6 if (i // 6) % 2 == 0 else 5 - i % 6 for i in range(2000)]
for batch in (1, 16):
    s = vs.stabilizer(p)
    if batch > 1:
        s.set_batch(batch); s.set_zero_copy(True)
    nb = max(4, 3 * batch)
    while len(outs) < nb:
        outs.append(capi.DevBuf(vs, fb))
    for i in range(96):
        s.push_dev(d_in.ptr + order[i] * fb, W, H, W, capi.FMT_NV12, outs[i % nb].ptr, W)
    s.sync()
    n = 320
    t0 = time.perf_counter()
    for i in range(96, 96 + n):
        s.push_dev(d_in.ptr + order[i] * fb, W, H, W, capi.FMT_NV12, outs[i % nb].ptr, W)
    s.sync()
    dt = time.perf_counter() - t0
    print("4K NV12 stabilize, batch %d: %.0f frames/s" % (batch, n / dt))
    s.close()
f = roll_scene.horizon_frame(W, H, 60, seed=1)
d_f, d_r, d_z = capi.DevBuf.from_array(vs, f), capi.DevBuf(vs, f.nbytes), capi.DevBuf(vs, f.nbytes)
rc, az = vs.roll_correction(), vs.auto_zoom_crop()
for _ in range(5):
    rc.correct_dev(d_f.ptr, W, H, W * 3, d_r.ptr, W * 3); rc.sync()
    az.apply_dev(d_r.ptr, W, H, W * 3, 3, d_z.ptr, W * 3); az.sync()
t0 = time.perf_counter()
for _ in range(50):
    rc.correct_dev(d_f.ptr, W, H, W * 3, d_r.ptr, W * 3)
rc.sync()
t1 = time.perf_counter()
for _ in range(50):
    az.apply_dev(d_r.ptr, W, H, W * 3, 3, d_z.ptr, W * 3)
az.sync()
t2 = time.perf_counter()
print("4K BGR roll correction: %.2f ms/frame; auto zoom/crop: %.2f ms/frame (info %s)" % ((t1 - t0) / 50 * 1e3, (t2 - t1) / 50 * 1e3, az.info().tolist()))
# the case the stage exists for: content rotated inside black borders (310-point contour, ~300 shrink rounds)
yy, xx = np.mgrid[:H, :W]
c, sn = np.cos(0.026), np.sin(0.026)
u = c * (xx - W / 2) + sn * (yy - H / 2) + W / 2 - 40
v = -sn * (xx - W / 2) + c * (yy - H / 2) + H / 2 + 25
f2 = np.maximum(f, 8)
f2[~((u >= 0) & (u < W) & (v >= 0) & (v < H))] = 0
d_f2 = capi.DevBuf.from_array(vs, f2)
for _ in range(5):
    az.apply_dev(d_f2.ptr, W, H, W * 3, 3, d_z.ptr, W * 3); az.sync()
t0 = time.perf_counter()
for _ in range(50):
    az.apply_dev(d_f2.ptr, W, H, W * 3, 3, d_z.ptr, W * 3)
az.sync()
print("4K BGR auto zoom/crop on rotated content: %.2f ms/frame (info %s)" % ((time.perf_counter() - t0) / 50 * 1e3, az.info().tolist()))
