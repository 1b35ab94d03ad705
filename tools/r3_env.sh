#!/bin/bash
# A/B of runtime environment knobs on the default bench (20 steps each)
o=gpurun_out/r3d; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$tag /" >> $o/env.txt; }
run base A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run nobranch RSSF_BRANCH_STREAMS=0
run nofork RSSF_FORK_FUSE=0
run devkernarg HIP_FORCE_DEV_KERNARG=1
run base2 A=1
cat $o/env.txt
