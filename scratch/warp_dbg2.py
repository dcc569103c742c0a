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
d_in = capi.DevBuf(vs, fb * B); d_out = capi.DevBuf(vs, fb * B + 65536)
d_out.zero()
M = np.zeros((B, 6), np.float32)
img = synth.render_frame(world, W, H, (300 * 256, 280 * 256, 90))
for b in range(B):
    d_in.upload(np.roll(img, 7 * b, axis=1), b * fb)
    ang = float(rng.normal(0, 0.002))
    M[b] = [np.cos(ang), -np.sin(ang), rng.normal(0, 3), np.sin(ang), np.cos(ang), rng.normal(0, 3)]
Mp = M.ctypes.data_as(C.POINTER(C.c_float))
print("d_in %x d_out %x" % (d_in.ptr, d_out.ptr))
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))
    vs.sync()
    rec = d_out.download((8 + 64 * 8,), np.uint64, offset=fb * B)
    n = int(rec[0] & 0xffffffff)
    if n:
        print("launch %d: %d bad tile inputs" % (it, n))
        for s in range(min(n, 12)):
            r = rec[8 + 8 * s: 16 + 8 * s]
            print("  wg %d  bz %d by %d bx %d  src %x dst %x  expected src %x  table %x" % (int(r[0]) & 0xffffffff, int(r[1]) >> 32, (int(r[1]) >> 16) & 0xffff, int(r[1]) & 0xffff, int(r[2]), int(r[3]), int(r[4]), int(r[5])))
        d_out.zero()
print("done")
