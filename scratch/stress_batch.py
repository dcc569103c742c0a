"""Soak of the batch schedule: random batch sizes, radii, stream lengths, smoothing methods, zero-copy on / off, BGR8 / NV12, pushes with
syncs at random places - the batch pipeline's outputs against the per-frame pipeline's (both on the GPU), for SECS seconds."""
import os, sys, time, random
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H = 320, 240
clips = {}
def clip_for(seed, nv12):
    k = (seed, nv12)
    if k not in clips:
        c = synth.make_clip(synth.SEED_CONFIG1 + seed, W, H, 24)
        clips[k] = [synth.bgr_to_nv12(f) for f in c] if nv12 else c
    return clips[k]
t0 = time.time(); runs = 0; frames = 0
while time.time() - t0 < SECS:
    nv12 = rng.random() < 0.3
    fmt = capi.FMT_NV12 if nv12 else capi.FMT_BGR8
    clip = clip_for(rng.randrange(4), nv12)
    fb = clip[0].nbytes; pitch = W if nv12 else W * 3
    batch = rng.choice([2, 3, 5, 8, 16, 27, 32, 33, 50, 64]); radius = rng.choice([2, 5, 9, 15, 30, 40]); n = rng.randrange(40, 260)
    method = rng.choice([0, 0, 1, 2])
    if method == 1: radius = min(radius, 15)          # (Gaussian kernels beyond sigma 10.5 are refused)
    zc = rng.random() < 0.5
    p = vs.params(smoothing_radius=radius, smoothing_method=method)
    s1, s2 = vs.stabilizer(p), vs.stabilizer(p)
    s2.set_batch(batch)
    if zc: s2.set_zero_copy(True)
    order = [i % 24 if (i // 24) % 2 == 0 else 23 - i % 24 for i in range(n)]
    d_in = capi.DevBuf(vs, fb * 24)
    for i, f in enumerate(clip): d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(vs, fb * (n + 4)), capi.DevBuf(vs, fb * (n + 4))
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + order[i] * fb, W, H, pitch, fmt, d_ref.ptr + k1 * fb, pitch)
        k2 += s2.push_dev(d_in.ptr + order[i] * fb, W, H, pitch, fmt, d_got.ptr + k2 * fb, pitch)
        if rng.random() < 0.02: s2.sync()
    while s1.flush_dev(d_ref.ptr + k1 * fb, pitch): k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, pitch): k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n, (k1, k2, n)
    shape = (k1, H * 3 // 2, W) if nv12 else (k1, H, W, 3)
    a, b = d_ref.download(shape, np.uint8), d_got.download(shape, np.uint8)
    if not np.array_equal(a, b):
        bad = [i for i in range(k1) if not np.array_equal(a[i], b[i])]
        print("MISMATCH: batch %d radius %d n %d method %d nv12 %s zero-copy %s: frames %s" % (batch, radius, n, method, nv12, zc, bad[:10])); sys.exit(1)
    for s in (s1, s2): s.close()
    for d in (d_in, d_ref, d_got): d.free()
    runs += 1; frames += n
    if runs % 10 == 0: print("%d runs, %d frames, %.0f s" % (runs, frames, time.time() - t0), flush=True)
print("ok: %d runs, %d frames, every output of the batch pipeline equal to the per-frame pipeline's" % (runs, frames))
