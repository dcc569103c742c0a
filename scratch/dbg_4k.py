import sys
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi, synth
import oracle_lib
vs = capi.load(); o = oracle_lib.load()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3, W, H, 8)]
o.lib.vso_set_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 8)
kw = dict(smoothing_radius=5, max_corners=400)
sg, so = vs.stabilizer(vs.params(**kw)), o.stabilizer(o.params(**kw))
for k, f in enumerate(clip):
    a, b = sg.push(f, 1), so.push(f, 1)
    if a is not None:
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        print("push", k, "max", d.max(), "nz", np.count_nonzero(d))
j = 0
while True:
    a, b = sg.flush(clip[0], 1), so.flush(clip[0], 1)
    if b is None: break
    dg, do = sg.debug(), so.debug()
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    ys, xs = np.nonzero(d)
    print("flush", j, "max", d.max(), "nz", len(ys), "rows", (ys.min(), ys.max()) if len(ys) else None,
          "M gpu", np.round(np.array(dg.warp_matrix), 6), "M ora", np.round(np.array(do.warp_matrix), 6), "idx", dg.out_index, do.out_index)
    j += 1
