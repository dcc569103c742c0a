import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np, ctypes as C
from vsamd import capi, synth
vs = capi.load()
W, H = 1920, 1080
world = synth.make_world(synth.SEED_CONFIG2, W, H)
img = synth.render_frame(world, W, H, (300 * 256, 280 * 256, 90))
fb = W * H * 3
for batch in (1, 16):
    d_in = capi.DevBuf(vs, fb * batch); d_out = capi.DevBuf(vs, fb * batch)
    for b in range(batch): d_in.upload(img, b * fb)
    M = np.tile(np.array([0.999998, -0.002, 2.75, 0.002, 0.999998, -1.25], np.float32), (batch, 1))
    Mp = M.ctypes.data_as(C.POINTER(C.c_float))
    for it in range(30):
        vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, batch, None))
    vs.sync()
    t0 = time.perf_counter()
    n = 200
    for it in range(n):
        vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, batch, None))
    vs.sync()
    dt = (time.perf_counter() - t0) / n
    print("batch %d: %.2f us per call, %.1f GB/s algorithmic" % (batch, dt * 1e6, 2 * fb * batch / dt / 1e9))
if len(sys.argv) > 1:
    import oracle_lib
    o = oracle_lib.load()
    out = d_out.download((H, W, 3), np.uint8)
    print("parity", np.array_equal(out, o.warp_affine(img, M[0], threads=8)))
