"""Phase stamps of the warp kernel (library built with -DWARP_ABL=9): mean shader cycles per phase over the interior tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
import numpy as np, ctypes as C
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
for it in range(5):
    vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))
vs.sync()
out = d_out.download((B, H, W * 3), np.uint8)
rows = []
for b in range(B):
    for by in range(H // 16):
        for bx in range(W // 128):
            q = out[b, by * 16, bx * 384: bx * 384 + 48].copy().view(np.uint64)
            if q[0] == 0x5354414d50:
                rows.append(q[1:6].astype(np.float64))
rows = np.array(rows)
names = ["entry -> corner terms (kernarg, table corners)", "-> staging loads landed, LDS written", "-> barrier passed", "-> rows emitted (blend, stores issued)", "-> stores acknowledged"]
print("%d stamped tiles of %d" % (len(rows), B * (H // 16) * (W // 128)))
for i, n in enumerate(names):
    print("  %-52s mean %8.0f  median %8.0f  p90 %8.0f cycles" % (n, rows[:, i].mean(), np.median(rows[:, i]), np.percentile(rows[:, i], 90)))
print("  total %.0f cycles" % rows.sum(axis=1).mean())
