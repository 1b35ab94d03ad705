#!/bin/bash
# HBM-side (fabric) traffic per kernel of a short command, from the L2's memory-side request counters (the derived
# FETCH_SIZE / WRITE_SIZE metrics did not finish on this image's rocprofv3).
# usage (on the GPU box through gpurun): tools/hbm_traffic.sh <tag> <command ...>   (keep the command to a few dozen launches:
# the derived counters replay every kernel several times; a whole training step does not finish in minutes)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=/tmp/hbm_$tag; mkdir -p $D gpurun_out
RSSF_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $D -o a -- "$@" > $D/a.log 2>&1
python - "$D" > gpurun_out/$tag.txt <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('rssf::', '')
        k = re.sub(r'\(.*', '', k)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'TCC_EA0_RDREQ_sum': cnt[k] += 1
print('# per launch.  read MB = TCC_EA0_RDREQ x 64 B x 2 (MI355X_MICROARCH.md, HBM: gfx950 tallies the 128-byte requests of wide')
print('# coalesced reads at 64 B); write MB = (WRREQ - WRREQ_64B) x 32 B + WRREQ_64B x 64 B (uncalibrated, as the guide says)')
print('%-64s %8s %12s %12s %10s %10s' % ('kernel', 'launches', 'RDREQ', 'WRREQ', 'read MB', 'write MB'))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['TCC_EA0_RDREQ_sum']):
    n = max(cnt[k], 1); rd, wr, w64 = d['TCC_EA0_RDREQ_sum'] / n, d['TCC_EA0_WRREQ_sum'] / n, d['TCC_EA0_WRREQ_64B_sum'] / n
    print('%-64s %8d %12.0f %12.0f %10.2f %10.2f' % (k[:64], cnt[k], rd, wr, rd * 128 / 1e6, ((wr - w64) * 32 + w64 * 64) / 1e6))
PY
tail -3 $D/a.log | cut -c1-300
head -40 gpurun_out/$tag.txt
