import csv, glob, sys, collections, re
tag, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'/root/repo/gpurun_out/{tag}/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r['Kernel_Name']):
            m = re.search(r'(\w+_kernel)(<[^>(]*>)?', r['Kernel_Name'])
            agg[m.group(0).replace('rssf::bf16_t', 'bf16') + ' grid=' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for n, v in sorted(d.items()):
        print('   %-24s %14.0f  (n=%d)' % (n, sum(v) / len(v), len(v)))
