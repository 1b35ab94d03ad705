#!/bin/bash
# NOTE: the variant this script switched on was measured and REMOVED from the code (DESIGN.md lesson 42, profiles/r04_stream_structure.txt): the variable is a no-op now; kept as the record of how the A/B was run.
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "RSSF_LOW0_SIDE=0" "RSSF_LOW0_SIDE=1" "RSSF_LOW0_SIDE=0" "RSSF_LOW0_SIDE=1" "RSSF_LOW0_SIDE=1 RSSF_LOCKSTEP_SPLIT=0,1"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>$o/err.txt | cut -c1-200 >> $o/bench.txt; tail -2 $o/err.txt | cut -c1-200; done
cat $o/bench.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_group.py -m gpu -x -q 2>&1 | tail -3
