#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RSSF_WGRAD_STREAM=0
for v in "A=1" "RSSF_LOCKSTEP_SPLIT=all" "RSSF_LOCKSTEP_SPLIT=all;all;0" "RSSF_LOCKSTEP_SPLIT=all;0;0" "RSSF_LOCKSTEP_SPLIT=0;all;all" "A=1"; do echo "== $v" >> $o/bench.txt; env "$v" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
