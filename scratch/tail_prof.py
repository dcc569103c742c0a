import sys, ctypes as C
sys.path.insert(0, 'video-stab_amd')
import numpy as np
from vsamd import capi, synth
vs = capi.load()
W, H = 1920, 1080
fb = W * H * 3
frames = synth.make_clip(synth.SEED_CONFIG2, W, H, 12)
d_in = capi.DevBuf(vs, fb * 12)
for i, f in enumerate(frames): d_in.upload(f, i * fb)
p = vs.params(smoothing_radius=30, max_corners=200, lk_win_size=21, lk_max_level=2)
s = vs.stabilizer(p); s.set_batch(16)
outs = [capi.DevBuf(vs, fb) for _ in range(32)]
order = [i % 12 if (i // 12) % 2 == 0 else 11 - i % 12 for i in range(2000)]
for i in range(161):
    s.push_dev(d_in.ptr + order[i] * fb, W, H, W * 3, 0, outs[i % 32].ptr, W * 3)
s.sync()
vs.lib.vs_stab_debug_ptr.restype = C.c_void_p
vs.lib.vs_stab_debug_ptr.argtypes = [C.c_void_p]
ptr = vs.lib.vs_stab_debug_ptr(s.h)
t = np.zeros(110, np.int64)
vs.check(vs.lib.vs_dev_memcpy_d2h(t.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), t.nbytes))
d = np.diff(t[:35]) / 100.0   # 100 MHz -> us
print("start->phase1 end: %.1f us" % d[0])
print("append:", np.round(d[1::2], 2))
print("emit  :", np.round(d[2::2], 2))

e = t[100:104]
print("emit parts (us): mirror+sync %.2f, atan+sync %.2f, wave0 %.2f" % tuple(np.diff(e) / 100.0))
