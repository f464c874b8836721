"""Per-kernel averages of SQ counters from a rocprofv3 --pmc csv (counter_collection.csv)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
names = sorted({c for v in acc.values() for c in v})
print('kernel,launches,' + ','.join(names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    n = max(cnt[k], 1)
    print('"%s",%d,' % (k, n) + ','.join('%.4g' % (v.get(c, 0)/n) for c in names))
