import csv, glob, sys, collections, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob((tag if tag.startswith('/') else f'/root/repo/gpurun_out/{tag}') + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)(<[^>(]*>)?', r['Kernel_Name'])
        if not m: continue
        k = m.group(0).replace('rssf::bf16_t', 'bf16').replace('rssf::wa::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
rows = []
for k, d in agg.items():
    wc = d.get('SQ_WAVE_CYCLES', 0)
    if wc <= 0: continue
    rows.append((wc, k, d))
print('%-52s %6s %6s %6s %6s %6s %7s %7s %6s' % ('kernel', 'wait%', 'istal%', 'ilds%', 'act%', 'valu%', 'valu/mf', 'lds/mf', 'bank%'))
for wc, k, d in sorted(rows, reverse=True)[:22]:
    mf = max(d.get('SQ_INSTS_MFMA', 0), 1)
    print('%-52s %6.1f %6.1f %6.1f %6.1f %6.1f %7.1f %7.2f %6.1f  hit%% %.0f' % (k[:52], 100 * d['SQ_WAIT_ANY'] / wc, 100 * d['SQ_WAIT_INST_ANY'] / wc, 100 * d.get('SQ_WAIT_INST_LDS', 0) / wc,
          100 * d['SQ_ACTIVE_INST_ANY'] / wc, 100 * d.get('SQ_ACTIVE_INST_VALU', 0) / wc, d['SQ_INSTS_VALU'] / mf, d['SQ_INSTS_LDS'] / mf,
          100 * d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1), 100 * d.get('TCC_HIT_sum', 0) / max(d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0), 1)))
