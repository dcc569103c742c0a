import sys
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi, synth
import oracle_lib
vs = capi.load(); o = oracle_lib.load()
img = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 1)[0]
for M in ([1,0,0,0,1,0], [1,0,5,0,1,-3], [0.99995,-0.01,3.25,0.01,0.99995,-7.5]):
    ref = o.warp_affine(img, np.array(M, np.float32))
    got = vs.warp_affine(img, M)
    bad = np.argwhere((ref != got).any(axis=2))
    print(M, "mismatch px:", len(bad))
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print(" rows", np.unique(ys)[:40], "cols", np.unique(xs)[:40], np.unique(xs)[-10:])
        for y, x in bad[:6]:
            print("  ", y, x, ref[y, x], got[y, x])
