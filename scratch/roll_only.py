import sys
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi
import roll_scene
vs = capi.load()
W, H = 3840, 2160
f = roll_scene.horizon_frame(W, H, 60, seed=1)
d_f, d_r = capi.DevBuf.from_array(vs, f), capi.DevBuf(vs, f.nbytes)
rc = vs.roll_correction()
for _ in range(30):
    rc.correct_dev(d_f.ptr, W, H, W * 3, d_r.ptr, W * 3)
rc.sync()
print(rc.state())
