#!/bin/bash
# usage: tools/prof_graph.sh <tag>   kernel trace of hipGraph-replayed training steps (default bench path)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r1 -- python bench.py --steps 3 --warmup 5 --no-cpu-baseline "$@" 2>&1 | tail -1 | grep -o "\"value.*ms_per_step[^,]*"
