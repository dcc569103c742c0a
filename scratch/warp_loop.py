"""warp_loop.py <frames> <launches> <sync each: 0/1>: repeated vs_op_warp_affine launches (VS_LIB picks the build), checks the output of the last one against the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
import numpy as np, ctypes as C
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
W, H = 1920, 1080
B, N, SY = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fb = W * H * 3
world = synth.make_world(synth.SEED_CONFIG2, W, H)
rng = np.random.default_rng(1)
d_in = capi.DevBuf(vs, fb * B); d_out = capi.DevBuf(vs, fb * B)
M = np.zeros((B, 6), np.float32)
img = synth.render_frame(world, W, H, (300 * 256, 280 * 256, 90))
for b in range(B):
    d_in.upload(np.roll(img, 7 * b, axis=1), b * fb)
    ang = float(rng.normal(0, 0.002))
    M[b] = [np.cos(ang), -np.sin(ang), rng.normal(0, 3), np.sin(ang), np.cos(ang), rng.normal(0, 3)]
Mp = M.ctypes.data_as(C.POINTER(C.c_float))
vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))
vs.sync()
ref = d_out.download((B, H, W, 3), np.uint8)
for it in range(N):
    vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))
    if SY:
        vs.sync()
vs.sync()
print("ok" if np.array_equal(ref, d_out.download((B, H, W, 3), np.uint8)) else "DIFFERENT", flush=True)
