#!/bin/bash
# SQ / TCC counter passes over a command; raw CSVs stay on the box (/tmp), only the per-kernel summary comes back
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=/tmp/pmc_$tag; mkdir -p $D gpurun_out
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $D -o a -- "$@" > $D/a.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $D -o b -- "$@" > $D/b.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D -o c -- "$@" > $D/c.log 2>&1
python tools/pmc_table.py $D > gpurun_out/$tag.txt 2>&1
tail -2 $D/a.log
