"""Timeline of ONE replayed (hipGraph) training step from a rocprofv3 kernel trace: per hardware queue the number of kernels, their
busy time and the idle gaps between them, and the longest kernels of the busiest queue (the critical chain of the step).
  on the GPU box:  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/<tag> -o r1 -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline
  then:            python tools/graph_timeline.py <tag>"""
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = sg[-2] + 1, sg[-1] + 1
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in step)
print("kernels %d, wall %.2f ms, sum of kernel durations %.2f ms" % (len(step), (t1 - t0) / 1e6, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6))
qkey = 'Queue_Id' if 'Queue_Id' in step[0] else 'Stream_Id'
byq = collections.defaultdict(list)
for r in step:
    byq[r[qkey]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
# union of busy intervals over all queues
iv = sorted((s, e) for r in byq.values() for s, e, _ in r)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("GPU busy (any queue) %.2f ms, idle %.2f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    ks.sort()
    dur = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print("queue %s: %4d kernels, busy %.2f ms, span %.2f ms, gaps < 20 us: %d, sum %.2f ms (median %.2f us); gaps >= 20 us: %d, sum %.2f ms; overlapping starts %d"
          % (q, len(ks), dur / 1e6, (ks[-1][1] - ks[0][0]) / 1e6, len(small), sum(small) / 1e6, (sorted(small)[len(small) // 2] / 1e3 if small else 0),
             sum(1 for g in gaps if g >= 20000), sum(g for g in gaps if g >= 20000) / 1e6, sum(1 for g in gaps if g < 0)))

# ---- where each queue waits: its gaps >= 30 us, the kernels around them and what the other queues ran meanwhile
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("rssf::", "").replace("cv::", "")
    return n.split('(')[0][:44]
allk = sorted((s, e, n, q) for q, ks in byq.items() for s, e, n in ks)
for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1]))[:3]:
    ks.sort()
    big = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1) if ks[i + 1][0] - ks[i][1] >= 30000]
    print("\nqueue %s: %d gaps >= 30 us, %.2f ms" % (q, len(big), sum(g for g, _ in big) / 1e6))
    for g, i in sorted(big, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
        gs, ge = ks[i][1], ks[i + 1][0]
        others = collections.Counter()
        for s, e, n, oq in allk:
            if oq != q and e > gs and s < ge:
                others[short(n)] += min(e, ge) - max(s, gs)
        top = ", ".join("%s %.0f" % (n, d / 1e3) for n, d in others.most_common(3))
        print("  %7.1f us at +%.2f ms  after %-40s before %-40s | meanwhile (us): %s" % (g / 1e3, (gs - t0) / 1e6, short(ks[i][2])[-40:], short(ks[i + 1][2])[-40:], top))
