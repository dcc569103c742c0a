"""A/B timing of vs_op_warp_affine builds, interleaved rounds in one process.

    python scratch/warp_ab.py [--frames 16] [--rounds 5] name=lib.so[:ENV=VAL,...] ...

Every variant is a (library, environment) pair; the environment is applied before the library is loaded
(a library reads VS_WARP_* once), so variants that differ only by environment need their own copy of the .so.
Each round runs every variant: 100 back-to-back launches, wall clock around them (the launches are GPU bound).
Prints the median and the minimum per variant, and checks every variant's output against the first one's.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
import numpy as np
import ctypes as C
from vsamd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=100)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

W, H, B = args.width, args.height, args.frames
fb = W * H * 3
tmpdir = tempfile.mkdtemp()
libs = []
for i, v in enumerate(args.variants):
    name, rest = v.split("=", 1)
    path, _, envs = rest.partition(":")
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition("=")
        os.environ[k] = val
    cp = os.path.join(tmpdir, "v%d.so" % i)       # a copy per variant: dlopen of one path twice gives one library
    shutil.copy(os.path.join(ROOT, path) if not os.path.isabs(path) else path, cp)
    libs.append((name, capi.VsLib(C.CDLL(cp)), [kv.partition("=")[0] for kv in filter(None, envs.split(","))], dict(kv.split("=", 1) for kv in filter(None, envs.split(",")))))
    for kv in filter(None, envs.split(",")):
        os.environ.pop(kv.partition("=")[0], None)

vs0 = libs[0][1]
_libs_env = libs
libs = [(n, v) for n, v, _, _ in libs]
world = synth.make_world(synth.SEED_CONFIG2, W, H)
rng = np.random.default_rng(1)
d_in = capi.DevBuf(vs0, fb * B)
d_out = capi.DevBuf(vs0, fb * B)
M = np.zeros((B, 6), np.float32)
for b in range(B):
    img = synth.render_frame(world, W, H, ((300 + 3 * b) * 256, (280 + 2 * b) * 256, 90 + 5 * b))
    d_in.upload(img, b * fb)
    ang = float(rng.normal(0, 0.002))
    M[b] = [np.cos(ang), -np.sin(ang), rng.normal(0, 3), np.sin(ang), np.cos(ang), rng.normal(0, 3)]
Mp = M.ctypes.data_as(C.POINTER(C.c_float))


def run(vs, n):
    for _ in range(n):
        vs.check(vs.lib.vs_op_warp_affine(d_in.ptr, W * 3, fb, d_out.ptr, W * 3, fb, W, H, 3, Mp, B, None))


ref = None
for (name, vs), (_, _, keys, env) in zip(libs, _libs_env):
    d_out.zero()
    os.environ.update(env)          # a library reads its VS_WARP_* settings at its first launch
    run(vs, 1)
    for k in keys:
        os.environ.pop(k, None)
    vs.sync()
    out = d_out.download((B, H, W, 3), np.uint8)
    if ref is None:
        ref = out
    else:
        print("%-12s output %s" % (name, "identical to " + libs[0][0] if np.array_equal(out, ref) else "DIFFERS from " + libs[0][0]))
times = {name: [] for name, _ in libs}
for r in range(args.rounds):
    for name, vs in libs:
        run(vs, 10)
        vs.sync()
        t0 = time.perf_counter()
        run(vs, args.launches)
        vs.sync()
        times[name].append((time.perf_counter() - t0) / args.launches * 1e6)
alg = 2.0 * fb * B
for name, _ in libs:
    t = sorted(times[name])
    med, mn = t[len(t) // 2], t[0]
    print("%-12s %d frames/launch: median %7.2f us  min %7.2f us  -> %6.1f GB/s algorithmic = %.3f of 8 TB/s (min: %.3f)" % (
        name, B, med, mn, alg / med / 1e3, alg / med / 1e3 / 8000.0, alg / mn / 1e3 / 8000.0))
shutil.rmtree(tmpdir, ignore_errors=True)
