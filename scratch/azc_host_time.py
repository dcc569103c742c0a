"""Time the host part of AutoZoomCrop (vs_azc_crop_from_mask) on a 4K mask: python scratch/azc_host_time.py"""
import sys, time
sys.path.insert(0, 'video-stab_amd')
import numpy as np
from vsamd import capi
vs = capi.load()
W, H = 3840, 2160
yy, xx = np.mgrid[:H, :W]
# a stabilised frame: content rotated by ~1.5 degrees and shifted, black outside
c, s = np.cos(0.026), np.sin(0.026)
u = c * (xx - W / 2) + s * (yy - H / 2) + W / 2 - 40
v = -s * (xx - W / 2) + c * (yy - H / 2) + H / 2 + 25
mask = (((u >= 0) & (u < W) & (v >= 0) & (v < H)) * 255).astype(np.uint8)
for _ in range(3):
    info = vs.azc_crop_from_mask(mask)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    info = vs.azc_crop_from_mask(mask)
print("host crop_from_mask: %.3f ms" % ((time.perf_counter() - t0) / n * 1e3), info)
