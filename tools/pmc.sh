#!/bin/bash
# usage: tools_pmc.sh <tag> <cmd...>  : SQ / LDS / TCC counter passes (kernel-trace only), outputs under gpurun_out/<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/$tag -o a -- "$@" > gpurun_out/$tag.a.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/$tag -o b -- "$@" > gpurun_out/$tag.b.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/$tag -o c -- "$@" > gpurun_out/$tag.c.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag -o d -- "$@" > gpurun_out/$tag.d.log 2>&1
tail -3 gpurun_out/$tag.a.log
