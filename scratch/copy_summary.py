"""Memory copies and kernels of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): python scratch/copy_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    by = collections.defaultdict(lambda: [0, 0, 0])
    for r in rows:
        k = r.get("Direction", r.get("Kind", "?"))
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        nb = int(r.get("Size", r.get("Bytes", 0)) or 0)
        big = "big" if nb >= (1 << 20) else "small"
        by[(k, big)][0] += 1; by[(k, big)][1] += dur; by[(k, big)][2] += nb
    for k, (n, t, b) in sorted(by.items()):
        print("  %-32s n %5d  mean %8.1f us  %8.1f MB  %6.2f GB/s while copying" % (k, n, t / n / 1e3, b / 1e6, b / max(t, 1)))
    if rows:
        t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
        print("  span %.2f ms, all copies %.1f MB" % ((t1 - t0) / 1e6, sum(int(r.get("Size", r.get("Bytes", 0)) or 0) for r in rows) / 1e6))
