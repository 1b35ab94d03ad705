#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o r1 -- python bench.py --steps 4 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python tools/graph_timeline.py $tag 40 > gpurun_out/$tag/timeline.txt 2>&1
find gpurun_out/$tag -name "*kernel_trace.csv" -delete; find gpurun_out/$tag -name "*.db" -delete
cat gpurun_out/$tag/timeline.txt | cut -c1-330
