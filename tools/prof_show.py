import csv, glob, sys, collections
tag = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
f = glob.glob(f'/root/repo/gpurun_out/{tag}/**/r1_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 5e6
print('total ms/step', round(tot, 1))
for r in rows[:n]:
    print(r['Name'][:72].ljust(72), r['Calls'].rjust(5), '%6.1f' % (float(r['TotalDurationNs']) / 5e6), '%7.1f' % (float(r['AverageNs']) / 1e3))
