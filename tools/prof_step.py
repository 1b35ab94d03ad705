import csv, glob, re, collections, sys
tag = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(f'/root/repo/gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = sg[-2] + 1, sg[-1] + 1
cnt = collections.Counter(); tim = collections.Counter()
def short(x):
    x = x.replace('void ', '').replace('(anonymous namespace)::', '').replace('rssf::', '').replace('at::native::', '')
    return x[:64]
for r in rows[a:b]:
    k = short(r['Kernel_Name']); cnt[k] += 1; tim[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('kernels in step', b - a, 'sum ms %.1f' % (sum(tim.values()) / 1e3), 'wall ms %.1f' % ((int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6))
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:n]:
    print('%6d %8.1f us  %s' % (cnt[k], v, k))
