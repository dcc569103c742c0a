import csv, sys, glob, re
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        m = re.search(r'(\w+_kernel(<[\w, ]+>)?|__amd\w+)', r['Name'])
        print("%-36s calls=%-6s avg_us=%-9.2f min_us=%-8.2f tot_ms=%-8.2f pct=%s" % ((m.group(1) if m else r['Name'][:36]), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
