"""Summary of a rocprofv3 --hip-trace run: per thread and per API call, count / total / mean duration; overlap of calls across threads.
python scratch/hip_api_summary.py <dir>"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Thread_Id"]), r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if not rows:
    print("no hip api trace found"); sys.exit(0)
t0 = min(r[2] for r in rows); t1 = max(r[3] for r in rows)
print("span %.1f ms, %d calls, %d threads" % ((t1 - t0) / 1e6, len(rows), len({r[0] for r in rows})))
per_thread = collections.defaultdict(lambda: [0, 0])
per_fn = collections.defaultdict(lambda: [0, 0])
for tid, fn, a, b in rows:
    per_thread[tid][0] += 1; per_thread[tid][1] += b - a
    per_fn[fn][0] += 1; per_fn[fn][1] += b - a
print("per thread: calls, ms inside HIP")
for tid, (n, d) in sorted(per_thread.items(), key=lambda kv: -kv[1][1]):
    print("  %8d %7d %9.2f" % (tid, n, d / 1e6))
print("per function: calls, total ms, mean us")
for fn, (n, d) in sorted(per_fn.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-34s %7d %9.2f %8.2f" % (fn, n, d / 1e6, d / n / 1e3))
# launches only: how long does a launch take when k launches of other threads are in flight at its start?
launch = [r for r in rows if "Launch" in r[1]]
ev = sorted([(a, 1) for _, _, a, b in launch] + [(b, -1) for _, _, a, b in launch])
import bisect
times = [e[0] for e in ev]; depth = []; d = 0
for e in ev:
    d += e[1]; depth.append(d)
by = collections.defaultdict(lambda: [0, 0])
for tid, fn, a, b in launch:
    k = depth[bisect.bisect_right(times, a) - 1]
    by[min(k, 6)][0] += 1; by[min(k, 6)][1] += b - a
print("kernel launches by number of launches in flight (all threads) at their start: count, mean us")
for k, (n, dd) in sorted(by.items()):
    print("  %d %7d %8.2f" % (k, n, dd / n / 1e3))
