"""pair_probe.py, the roll stage only: alone and beside the stabilizer (frames/s), plus how many frames took the slow path."""
import os, sys, time, threading
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from vsamd import capi, synth
vs = capi.load(os.environ.get("VS_LIB"))
W, H, NF, CH = 3840, 2160, 64, 128
sb = W * H * 3 // 2
clip = synth.make_clip_dev(vs, synth.SEED_CONFIG3, W, H, NF, nv12=True)
srcs = [clip.ptr + (i % NF) * sb for i in range(CH)]
outs = {k: capi.DevBuf(vs, sb * CH) for k in "RS"}
rc = vs.roll_correction()
st = vs.stabilizer(bench.make_params(vs, max_corners=400), device=0)
st.set_batch(64); st.set_zero_copy(True)
def roll():
    rc.correct_nv12_dev_n(srcs, W, H, W, [outs["R"].ptr + i * sb for i in range(CH)], W); rc.sync(); return CH
def stab():
    st.push_dev_n(srcs, W, H, W, capi.FMT_NV12, [outs["S"].ptr + i * sb for i in range(CH)], W); st.sync(); return CH
F = {"R": roll, "S": stab}
for f in F.values():
    for _ in range(3): f()
def run(combo, secs=0.6):
    stop = [False]; res = {}
    def loop(k):
        n = 0; t0 = time.perf_counter()
        while not stop[0]: n += F[k]()
        res[k] = n / (time.perf_counter() - t0)
    ths = [threading.Thread(target=loop, args=(k,)) for k in combo]
    for t in ths: t.start()
    time.sleep(secs); stop[0] = True
    for t in ths: t.join()
    return res
for combo in ("R", "RS", "R", "RS"):
    r = run(combo)
    print("%-4s" % combo, "  ".join("%s %6.0f" % (k, r[k]) for k in combo), flush=True)
