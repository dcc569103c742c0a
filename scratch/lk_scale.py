"""lk_kernel time vs number of points in one launch (is the batched tracking latency- or issue-bound?)."""
import sys, time, ctypes as C
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi, synth
import oracle_lib
vs = capi.load(); o = oracle_lib.load()
clip = synth.make_clip(synth.SEED_CONFIG2, 1920, 1080, 2)
g0 = o.analysis_gray(clip[0], 960, 540); g1 = o.analysis_gray(clip[1], 960, 540)
pts, _ = o.gftt(g0, 200, 0.02, 15.0, 3)
h, w = g0.shape
# pyramids + derivatives built once by the op; time only repeated calls of the whole op minus a 1-point call
d_prev, d_next = capi.DevBuf.from_array(vs, g0), capi.DevBuf.from_array(vs, g1)
for n in (1, 200, 800, 3200, 6400):
    P = np.tile(pts, ((n + len(pts) - 1) // len(pts), 1))[:n].astype(np.float32)
    d_pts = capi.DevBuf.from_array(vs, P)
    d_out, d_st, d_err = capi.DevBuf(vs, n * 8), capi.DevBuf(vs, n), capi.DevBuf(vs, n * 4)
    def call():
        vs.check(vs.lib.vs_op_pyr_lk(d_prev.ptr, d_next.ptr, w, w, h, d_pts.ptr, n, d_out.ptr, d_st.ptr, d_err.ptr, 21, 2, 20, 0.03, None))
    for _ in range(3): call()
    vs.sync()
    t0 = time.perf_counter()
    for _ in range(10): call()
    vs.sync()
    print("n=%5d: %.1f us per op call (pyramids + scharr + LK, synchronous op)" % (n, (time.perf_counter() - t0) / 10 * 1e6))
