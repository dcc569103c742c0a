"""pmc_per_wave.py <rocprof dir> [filter]: counters of the largest dispatches of each kernel, per wave (SQ_WAVES)."""
import csv, glob, sys, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ''
vals = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace('vsd::(anonymous namespace)::', '').split('(')[0]
        if flt in k:
            d = int(r["Dispatch_Id"])
            vals[k][r["Counter_Name"]][d] = vals[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
for k, cs in sorted(vals.items()):
    w = cs.get("SQ_WAVES", {})
    if not w:
        continue
    top = max(w.values())
    ids = [d for d, v in w.items() if v == top]
    line = "  %-34s waves %8d x%3d:" % (k[:34], top, len(ids))
    for c in sorted(cs):
        if c != "SQ_WAVES":
            line += " %s %.1f" % (c.replace("SQ_", "").replace("INSTS_", ""), sum(cs[c][d] for d in ids if d in cs[c]) / (top * len(ids)))
    print(line)
