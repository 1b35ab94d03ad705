#!/bin/bash
# usage: tools_prof.sh <tag> [extra bench args]   (run on the GPU box through gpurun)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RSSF_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | grep -o "\"value.*ms_per_step[^,]*"
