"""Per-(kernel, grid) durations of one eager step: python tools/prof_launches.py <gpurun_out tag> [substring ...]"""
import csv, glob, collections, sys
tag = sys.argv[1]; subs = sys.argv[2:]
f = glob.glob(f'gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = sg[-2] + 1, sg[-1] + 1
agg = collections.defaultdict(list)
for r in rows[a:b]:
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('rssf::', '')[:56]
    if subs and not any(s in k for s in subs):
        continue
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    agg[(k, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%4d x %8.1f us = %8.1f us  blocks %6d  %s' % (len(v), sum(v) / len(v), sum(v), g, k))
