"""Time of the pyramid-level kernel alone: 32 images of 960x540 per launch (level 0 of a batch), then levels 1 and 2."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'video-stab_amd'))
import numpy as np
from vsamd import capi
vs = capi.load()
for (w, h) in ((960, 540), (480, 270), (240, 135)):
    n = 32
    g = np.random.default_rng(0).integers(0, 256, (n, h, w), dtype=np.uint8)
    d_in = capi.DevBuf.from_array(vs, g)
    d_der = capi.DevBuf(vs, n * h * w * 4)
    dw, dh = (w + 1) // 2, (h + 1) // 2
    d_next = capi.DevBuf(vs, n * dw * dh)
    for down in (True, False):
        def run():
            vs.check(vs.lib.vs_op_pyr_level(d_in.ptr, w, w * h, w, h, d_der.ptr, d_next.ptr if down else None, dw, dw * dh, n, None))
        for _ in range(3): run()
        t0 = time.perf_counter()
        for _ in range(20): run()
        print("%dx%d x%d down=%d: %.1f us per call (incl. table upload + sync)" % (w, h, n, down, (time.perf_counter() - t0) / 20 * 1e6))
