import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
import numpy as np, ctypes as C
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
W, H, B = 1920, 1080, 32
fb = W * H * 3
world = synth.make_world(synth.SEED_CONFIG2, W, H)
rng = np.random.default_rng(1)
NWG = 1024
d_in = capi.DevBuf(vs, fb * B); d_out = capi.DevBuf(vs, fb * B + NWG * 128)
M = np.zeros((B, 6), np.float32)
img = synth.render_frame(world, W, H, (300 * 256, 280 * 256, 90))
for b in range(B):
    d_in.upload(np.roll(img, 7 * b, axis=1), b * fb)
    ang = float(rng.normal(0, 0.002))
    M[b] = [np.cos(ang), -np.sin(ang), rng.normal(0, 3), np.sin(ang), np.cos(ang), rng.normal(0, 3)]
Mp = M.ctypes.data_as(C.POINTER(C.c_float))
for it in range(5):
    vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))
vs.sync()
r = d_out.download((NWG, 16), np.uint64, offset=fb * B).astype(np.float64)
ni, ne = r[:, 6].sum(), r[:, 7].sum()
print("tiles: %d interior, %d other; per workgroup: total cycles mean %.0f  min %.0f  max %.0f" % (ni, ne, r[:, 8].mean(), r[:, 8].min(), r[:, 8].max()))
print("interior tiles: to LDS (waits for its loads) %.0f   emit %.0f cycles per tile" % (r[:, 0].sum() / ni, r[:, 3].sum() / ni))
print("other tiles:    staging %.0f   emit %.0f cycles per tile" % (r[:, 4].sum() / max(ne, 1), r[:, 5].sum() / max(ne, 1)))
print("issue of the next tile: interior %.0f   other %.0f cycles;   barrier %.0f cycles per tile" % (r[:, 1].sum() / max(ni + ne - r[:, 10].sum(), 1), r[:, 9].sum() / max(r[:, 10].sum(), 1), r[:, 2].sum() / (ni + ne)))
w = np.argsort(r[:, 8])
print("slowest workgroups:", [(int(i), int(r[i, 8]), int(r[i, 6]), int(r[i, 7])) for i in w[-5:]])
print("fastest workgroups:", [(int(i), int(r[i, 8]), int(r[i, 6]), int(r[i, 7])) for i in w[:5]])
