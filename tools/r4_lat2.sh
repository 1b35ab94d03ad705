#!/bin/bash
# lattice kernel diagnosis: batch sweep (rounds of workgroups) and SQ / LDS / TCC counters of both 17-tap kernels
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 7 14 16; do echo "B=$b" >> $o/sweep.txt; timeout 200 python tools/lattice_bench.py $b >> $o/sweep.txt 2>&1; done; grep -v amdgpu.ids $o/sweep.txt
tools/pmc_full.sh ${1}_pmc python tools/lattice_bench.py 16 > /dev/null 2>&1
D=/tmp/pmc_x; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $D -o d -- python tools/lattice_bench.py 16 > $D/d.log 2>&1
python - $D > $o/pmc_extra.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'conv_' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k] += 1
for k, d in agg.items():
    n = max(cnt[k], 1)
    print(k, 'launches', n, {c: round(v / n) for c, v in d.items()})
PY
cat gpurun_out/${1}_pmc.txt | head -12 | cut -c1-200; cat $o/pmc_extra.txt; tail -3 $D/d.log
