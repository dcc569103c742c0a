"""How fast can the host move one 1080p frame (6.2 MB) between two buffers with k threads, and what do the four
combinations of (pageable | page-locked) x (synchronous | host pipeline) of vs_stab_push deliver?  (scratch, GPU box)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-stab_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

def copy_rate(k, n=200, fresh=False):
    src = [np.random.randint(0, 255, (1080, 1920, 3), dtype=np.uint8) for _ in range(4)]
    dst = np.empty_like(src[0])
    rows = np.array_split(np.arange(1080), k)
    def part(d, s, r): np.copyto(d[r[0]:r[-1] + 1], s[r[0]:r[-1] + 1])
    t0 = time.perf_counter()
    for i in range(n):
        if fresh: dst = np.empty_like(src[0])
        th = [threading.Thread(target=part, args=(dst, src[i % 4], r)) for r in rows[1:]]
        for t in th: t.start()
        part(dst, src[i % 4], rows[0])
        for t in th: t.join()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e3, 6.2208e-3 / dt

print("host cores", os.cpu_count())
for k in () if os.environ.get("VS_RATE_ONLY_PUSH") else (1, 2, 4, 8, 16):
    ms, gbs = copy_rate(k)
    print("memcpy of a 1080p frame, %2d threads: %.3f ms  %.1f GB/s" % (k, ms, gbs))
if not os.environ.get("VS_RATE_ONLY_PUSH"):
    ms, gbs = copy_rate(1, fresh=True)
    print("  into a fresh np.empty each time, 1 thread: %.3f ms" % ms)

import bench
from vsamd import capi, synth
vs = capi.load()
params = bench.make_params(vs)
frames = synth.make_clip(synth.SEED_CONFIG2, 1920, 1080, 12)
order = bench.clip_order(len(frames), 64 + 240)

def run(pinned, pipeline, fresh_out):
    s = vs.stabilizer(params, device=0)
    if pipeline: s.set_host_pipeline(True)
    src, out, bufs = frames, None, []
    if pinned:
        bufs = [capi.HostBuf(vs, f.shape) for f in frames] + [capi.HostBuf(vs, frames[0].shape)]
        for b, f in zip(bufs, frames): b.array[...] = f
        src, out = [b.array for b in bufs[:-1]], bufs[-1].array
    elif not fresh_out:
        out = np.empty_like(frames[0])
    for i in order[:64]: s.push(src[i], out=out)
    t0 = time.perf_counter(); n = 0
    for i in order[64:]:
        if s.push(src[i], out=out) is not None: n += 1
    dt = time.perf_counter() - t0
    s.close()
    for b in bufs: b.free()
    return n / dt

modes = [(0, 0, 1), (0, 0, 0), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 1, 0)]
if os.environ.get("VS_RATE_ONLY_PUSH"): modes = [(0, 0, 0), (0, 1, 1), (0, 1, 0)]
print("VS_STAB_HOST_HELPER", os.environ.get("VS_STAB_HOST_HELPER"))
for pinned, pipeline, fresh in modes:
    print("vs_stab_push  %-11s %-13s %-22s %7.1f frames/s" % ("page-locked" if pinned else "pageable", "host pipeline" if pipeline else "synchronous",
          "fresh output per call" if fresh else "one output buffer", run(pinned, pipeline, fresh)))
