#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "A=1" "RSSF_BM256_MINK=448" "RSSF_BM256_MINK=256" "A=1" "RSSF_BM256_MINK=448"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
RSSF_BM256_MINK=448 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
