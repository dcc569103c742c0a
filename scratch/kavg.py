"""kavg.py <rocprof dir> [filter]: average / min / max duration per kernel name from *_kernel_stats.csv."""
import csv, glob, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if flt in r['Name']:
            print("  %-60s calls %5s avg %9.1f us min %8.1f max %8.1f" % (r['Name'].replace('vsd::(anonymous namespace)::', '')[:60], r['Calls'],
                  float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
