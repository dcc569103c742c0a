"""timeline.py <rocprofv3 dir>: kernels and copies of the last steady-state batches of a `bench.py` run, with start offsets
(us) relative to a tail kernel, durations, queue and stream ids (from --kernel-trace --memory-copy-trace CSVs)."""
import csv, glob, re, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel|__amd_rocclr_\w+)', r['Kernel_Name'])
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:30], r['Queue_Id'], r['Stream_Id']))
for f in glob.glob(d + '/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY', 'c', r.get('Stream_Id', '')))
ev.sort()
tails = [i for i, e in enumerate(ev) if e[2] == 'ransac_tail_batch_kernel']
print('tail-to-tail periods (us):', [round((ev[tails[k + 1]][0] - ev[tails[k]][0]) / 1e3, 1) for k in range(max(0, len(tails) - 5), len(tails) - 1)])
i0, i1 = tails[-4], tails[-3]
t0 = ev[i0][0]
for e in ev[i0:i1 + 1]:
    print("%9.1f %8.1f  q%s s%s  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[4], e[2]))
