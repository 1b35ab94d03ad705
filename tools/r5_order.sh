#!/bin/bash
tag=$1; o=gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_order.py $tag $2 $3 > $o/order.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete
wc -l $o/order.txt
