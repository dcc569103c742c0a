"""kavg.py <rocprof dir>: average duration per kernel name from *_kernel_stats.csv (first 60 chars of the name)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'enh_' in r['Name'] or 'nlm_' in r['Name']:
            print("  %-60s calls %5s avg %9.1f ns min %8s" % (r['Name'].replace('vsd::(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']), r['MinNs']))
