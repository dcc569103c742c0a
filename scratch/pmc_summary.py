import csv, collections, re, sys, glob
d = sys.argv[1]
kt = {}
for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        kt[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp'])-int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
seen=set()
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        m = re.search(r'(\w+_kernel(<\d+(, \w+)?>)?)', name)
        short = m.group(1) if m else name[:30]
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
        key=(short,r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            dd = kt.get(r['Dispatch_Id'])
            if dd: agg[short]['dur_ns'].append(dd[1])
for k,v in agg.items():
    line = "%-26s n=%d" % (k, len(v['dur_ns']))
    for c in sorted(v):
        line += " %s=%.0f" % (c.replace('SQ_','').replace('GRBM_','').replace('INSTS_',''), sum(v[c])/len(v[c]))
    print(line)
