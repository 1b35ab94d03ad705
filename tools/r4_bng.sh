#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
for v in "RSSF_BN_GROUP=4" "RSSF_BN_GROUP=12" "RSSF_BN_GROUP=4" "RSSF_BN_GROUP=12"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
