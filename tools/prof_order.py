"""Kernels of one eager step in launch order with durations: python tools/prof_order.py <gpurun_out tag> <first n> <last n>"""
import csv, glob, sys
tag, n0, n1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(f'gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = sg[-2] + 1, sg[-1] + 1
step = rows[a:b]
def line(i, r):
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('rssf::', '')[:70]
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    return '%4d %8.1f us  blocks %6d  %s' % (i, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, g, k)
for i, r in enumerate(step[:n0]): print(line(i, r))
print('...')
for i, r in enumerate(step[-n1:]): print(line(len(step) - n1 + i, r))
