"""Summarise FETCH_SIZE / WRITE_SIZE per kernel from rocprofv3 --pmc csv output (counter_collection.csv).
usage: pmc_traffic.py <fetch_csv> <write_csv>"""
import csv, sys, collections
def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter: continue
            a = acc[r['Kernel_Name']]
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return acc
fe = load(sys.argv[1], 'FETCH_SIZE'); wr = load(sys.argv[2], 'WRITE_SIZE')
print('kernel,launches,FETCH_SIZE_raw_per_launch,WRITE_SIZE_raw_per_launch')
for k in sorted(fe, key=lambda k: -fe[k][1]):
    n = fe[k][0]
    print('"%s",%d,%.1f,%.1f' % (k[:90], n, fe[k][1]/n, wr[k][1]/max(wr[k][0], 1) if k in wr else float('nan')))
