"""Enhancer timing on the GPU box: frames resident in HBM, host clock around a stream sync.
python scratch/enh_bench.py [W H] ; prints one line per configuration."""
import sys, time
sys.path.insert(0, 'video-stab_amd'); sys.path.insert(0, 'tests')
import numpy as np
from vsamd import capi
from test_enhance import scene, CONFIGS
vs = capi.load()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
NB = 16
frames = [scene(W, H, seed=s) for s in range(2)]
ins = [capi.DevBuf.from_array(vs, frames[i % 2]) for i in range(NB)]
outs = [capi.DevBuf(vs, W * H * 3) for _ in range(NB)]
e = capi.Enhancer(vs)
for name in ("shipped", "cb_only", "vibrance_only", "wb_only", "clahe_only", "unsharp_wide", "all_cpu_order", "all_cuda_order", "denoise_only"):
    p = capi.Enhancer.default_params(vs, **CONFIGS[name])
    def run(iters, batch):
        for _ in range(iters):
            if batch:
                e.apply_batch_dev(p, [b.ptr for b in ins], [b.ptr for b in outs], W, H, W * 3, W * 3)
            else:
                for i in range(NB):
                    e.apply_dev(p, ins[i].ptr, W, H, W * 3, outs[i].ptr, W * 3)
        e.sync()
    res = []
    for batch in (0, 1):
        run(1 if name.startswith('denoise') else 3, batch)
        iters = 2 if name.startswith('denoise') else 20
        t0 = time.perf_counter(); run(iters, batch); dt = time.perf_counter() - t0
        res.append(dt / (iters * NB) * 1e6)
    print("%-16s %dx%d passes=%d  single %.1f us/frame  batch16 %.1f us/frame  (%.1f GB/s algorithmic at 6 B/px)" % (
        name, W, H, e.passes(), res[0], res[1], W * H * 6 / (res[1] * 1e-6) / 1e9), flush=True)
