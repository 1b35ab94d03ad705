"""What runs on the OTHER hardware queues while a given kernel runs inside a replayed (hipGraph) step: for every launch of the kernels whose
name contains <pattern> in the last replayed steps of a rocprofv3 kernel trace, the overlapping kernels of other queues, summed (us of overlap).
  on the GPU box:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/<tag> -o r1 -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline [--size 128 --batch 2]
  then:            python tools/graph_overlap.py <tag> <pattern>"""
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f'gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
rows = rows[sg[-6] + 1:sg[-1] + 1]                   # the last five steps
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("rssf::", "").replace("cv::", "").split('(')[0][:70]
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r[qkey]) for r in rows]
mine = [k for k in ks if pat in k[2]]
print("%d launches of %s in 5 steps, %.1f us each" % (len(mine), sorted({k[2] for k in mine}), sum(e - s for s, e, _, _ in mine) / 1e3 / max(1, len(mine))))
tot, alone = collections.Counter(), 0
for s, e, n, q in mine:
    ov = [(min(e, e2) - max(s, s2), n2) for s2, e2, n2, q2 in ks if q2 != q and e2 > s and s2 < e]
    if not ov:
        alone += 1
    for d, n2 in ov:
        tot[n2] += d
print("launches with nothing on another queue: %d" % alone)
for n2, d in tot.most_common(25):
    print("  %9.1f us  %s" % (d / 1e3, n2))
