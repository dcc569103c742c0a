import sys
sys.path.insert(0, 'video-stab_amd')
import numpy as np, ctypes as C
from vsamd import capi, synth
import os
vs = capi.load(os.environ.get("VS_LIB"))
W, H = 1920, 1080
img = synth.render_frame(synth.make_world(synth.SEED_CONFIG2, W, H), W, H, (300 * 256, 280 * 256, 90))
fb = W * H * 3; batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d_in = capi.DevBuf(vs, fb * batch); d_out = capi.DevBuf(vs, fb * batch)
for b in range(batch): d_in.upload(img, b * fb)
M = np.tile(np.array([0.999998, -0.002, 2.75, 0.002, 0.999998, -1.25], np.float32), (batch, 1))
Mp = M.ctypes.data_as(C.POINTER(C.c_float))
for it in range(20):
    vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, batch, None))
vs.sync()
