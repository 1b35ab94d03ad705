#!/bin/bash
# raw counter sums per kernel (averaged per launch) for a command: tools/pmc_raw.sh <tag> <kernel-substring> <cmd...>
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=/tmp/pmcraw_$tag; mkdir -p $D gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D -o a -- "$@" > $D/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $D -o b -- "$@" > $D/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $D -o c -- "$@" > $D/c.log 2>&1
python - "$D" "$pat" <<'PY' > gpurun_out/$tag.txt 2>&1
import csv, glob, sys, collections
D, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(D + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(agg): print('%-34s %16.0f per launch (%d launches)' % (k, agg[k] / n[k], n[k]))
PY
cat gpurun_out/$tag.txt
