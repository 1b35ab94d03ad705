#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "A=1" "RSSF_HALO_NTB=2" "RSSF_LATTICE=1" "RSSF_WGRAD_T2=0" "RSSF_HALO_NTB=1" "A=1" "RSSF_GROUP_WGRAD_TPB=8" "RSSF_FORK_FUSE=0"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
