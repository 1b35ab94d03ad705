#!/bin/bash
# eager kernel trace of one step, then per-(kernel, grid) durations: tools/r5_launches.sh <tag> "<substrings>" [VAR=val ...]
tag=$1; subs=$2; shift 2
o=gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
env "$@" RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_launches.py $tag $subs > $o/launches.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
head -60 $o/launches.txt
