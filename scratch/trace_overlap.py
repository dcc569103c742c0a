"""Kernel-trace summary: per kernel group (roll / stab / zoom) busy time and pairwise overlap in a window (rocprofv3 --kernel-trace csv)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
def group(n):
    if any(k in n for k in ("canny", "hough", "sobel", "edge_list", "resize_gray_kernel<1")): return "roll"
    if any(k in n for k in ("threshold_bits", "close5")): return "zoom"
    if "warp_nv12_kernel<3>" in n or "warp_nv12_kernelILi3" in n: return "roll"
    if "warp_affine_kernel" in n: return "zoom"
    if "copyBuffer" in n or "fillBuffer" in n: return "copy"
    return "stab"
t_end = rows[-1][1]
win = [r for r in rows if r[0] > t_end - 60_000_000]      # the last 60 ms
t0, t1 = win[0][0], win[-1][1]
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
ev = []
for s, e, n, q in win:
    g = group(n)
    ev.append((s, 1, g)); ev.append((e, -1, g))
ev.sort()
act = {"roll": 0, "stab": 0, "zoom": 0, "copy": 0}
busy = {k: 0 for k in act}
anyb = 0; multi = 0
last = t0
for t, d, g in ev:
    dt = t - last
    n_act = sum(1 for k in ("roll", "stab", "zoom") if act[k] > 0)
    if n_act >= 1 or act["copy"] > 0: anyb += dt
    if n_act >= 2: multi += dt
    for k in act:
        if act[k] > 0: busy[k] += dt
    act[g] += d
    last = t
tot = t1 - t0
print("GPU busy (any kernel) %.1f %%; two or more stages at once %.1f %%" % (100.0 * anyb / tot, 100.0 * multi / tot))
for k in busy: print("  %-5s active %.1f %% of the window" % (k, 100.0 * busy[k] / tot))
qs = {}
for s, e, n, q in win: qs.setdefault((group(n), q), 0); qs[(group(n), q)] += 1
print("kernels per (stage, queue):", sorted(qs.items()))
