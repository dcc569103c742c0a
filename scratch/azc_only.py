"""30 calls of vs_azc_apply_dev on a 4K frame (for rocprofv3) + wall time: python scratch/azc_only.py [rotated]"""
import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi
import roll_scene
vs = capi.load()
W, H = 3840, 2160
f = np.maximum(roll_scene.horizon_frame(W, H, 60, seed=1), 8)
if len(sys.argv) > 1:
    yy, xx = np.mgrid[:H, :W]
    c, s = np.cos(0.026), np.sin(0.026)
    u = c * (xx - W / 2) + s * (yy - H / 2) + W / 2 - 40
    v = -s * (xx - W / 2) + c * (yy - H / 2) + H / 2 + 25
    f[~((u >= 0) & (u < W) & (v >= 0) & (v < H))] = 0
d_f, d_z = capi.DevBuf.from_array(vs, f), capi.DevBuf(vs, f.nbytes)
az = vs.auto_zoom_crop()
for _ in range(5):
    az.apply_dev(d_f.ptr, W, H, W * 3, 3, d_z.ptr, W * 3); az.sync()
t0 = time.perf_counter()
for _ in range(30):
    az.apply_dev(d_f.ptr, W, H, W * 3, 3, d_z.ptr, W * 3)
az.sync()
print("azc %.3f ms/frame, info %s" % ((time.perf_counter() - t0) / 30 * 1e3, az.info().tolist()))
