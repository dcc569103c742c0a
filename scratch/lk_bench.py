import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np, ctypes as C
from vsamd import capi, synth
import oracle_lib
vs = capi.load(); o = oracle_lib.load()
clip = synth.make_clip(synth.SEED_CONFIG2, 1920, 1080, 2)
g0 = o.analysis_gray(clip[0], 960, 540); g1 = o.analysis_gray(clip[1], 960, 540)
pts, _ = o.gftt(g0, 200, 0.02, 15.0, 3)
print("npts", len(pts))
no, so, eo = o.pyr_lk(g0, g1, pts, 21, 2, 20, 0.03)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    ng, sg, eg = vs.pyr_lk(g0, g1, pts, 21, 2, 20, 0.03)
print("match", np.array_equal(sg, so), np.array_equal(ng.view(np.uint32), no.view(np.uint32)))
pg = vs.gftt(g0, 200, 0.02, 15.0, 3)
for rep in range(10): pg = vs.gftt(g0, 200, 0.02, 15.0, 3)
print("gftt match", np.array_equal(pg, pts))
