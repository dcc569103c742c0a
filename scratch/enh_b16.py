"""20 x 16 frames through the enhancer (argv: config name, "single" for one launch per frame instead of one per
16 frames), for rocprofv3 --stats / --pmc runs."""
import sys
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
from vsamd import capi
from test_enhance import scene, CONFIGS
vs = capi.load()
W, H, NB = 1920, 1080, 16
cfg = sys.argv[1] if len(sys.argv) > 1 else "shipped"
frames = [scene(W, H, seed=s) for s in range(2)]
ins = [capi.DevBuf.from_array(vs, frames[i % 2]) for i in range(NB)]
outs = [capi.DevBuf(vs, W * H * 3) for _ in range(NB)]
e = capi.Enhancer(vs)
p = capi.Enhancer.default_params(vs, **CONFIGS[cfg])
single = len(sys.argv) > 2 and sys.argv[2] == "single"
for _ in range(20):
    if single:
        for i in range(NB):
            e.apply_dev(p, ins[i].ptr, W, H, W * 3, outs[i].ptr, W * 3)
    else:
        e.apply_batch_dev(p, [b.ptr for b in ins], [b.ptr for b in outs], W, H, W * 3, W * 3)
e.sync()
