import sys
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi, synth
vs = capi.load()
clip = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 12)
s = vs.stabilizer(vs.params(smoothing_radius=8))
for k, f in enumerate(clip):
    print("push", k, flush=True)
    o = s.push(f)
    print("  ->", o is not None, flush=True)
    d = s.debug()
    print("  dbg", d.n_prev, d.n_valid, d.n_inliers, list(d.transform), flush=True)
