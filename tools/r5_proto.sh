#!/bin/bash
# round 5: A-direct MLP convolution prototype (tools/mlp_direct_proto.hip) against the shipping gather kernel, then the step baseline
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 tools/bin/mlp_proto 16 > $o/proto.txt 2>&1; echo "rc $?" >> $o/proto.txt
cat $o/proto.txt
if [ "$2" = "bench" ]; then
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; tail -c 1500 $o/bench.json
fi
