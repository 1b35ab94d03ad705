#!/bin/bash
# usage: tools/r3_ab.sh <tag> "ENV=.. ENV=.." "ENV=.." ...   default bench (30 steps) under each environment, twice each interleaved
o=gpurun_out/$1; shift; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  r=$(env $v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$v  $r" >> $o/ab.txt
done; done
cat $o/ab.txt
