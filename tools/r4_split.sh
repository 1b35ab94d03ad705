#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "0" "0;0;0,1" "0;0,1;0" "0,3" "0;0;0,3" "0;0,2;0,3" "1" "0;0;1" "0"; do echo "== RSSF_LOCKSTEP_SPLIT=$v" >> $o/bench.txt; RSSF_LOCKSTEP_SPLIT="$v" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
