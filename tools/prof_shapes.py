import csv, glob, collections, re, sys
tag = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else 'conv'
f = glob.glob(f'/root/repo/gpurun_out/{tag}/**/r1_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if re.search(pat, n):
        m = re.search(r'(\w+_kernel)(<[^>(]*>)?', n)
        d[(m.group(0).replace('rssf::bf16_t', 'bf16'), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(k, len(v) // 5, 'avg %.1f' % (sum(v) / len(v)), 'tot/step %.2f ms' % (sum(v) / 5e3))
