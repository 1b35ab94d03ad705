#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "A=1" "RSSF_WGRAD_HALO_TPB=16 RSSF_WGRAD_HALO_MINBLK=128" "RSSF_WGRAD_HALO_TPB=16 RSSF_WGRAD_HALO_MINBLK=256" "RSSF_WGRAD_HALO_TPB=32 RSSF_WGRAD_HALO_MINBLK=64" "RSSF_GROUP_WGRAD_TPB=32 RSSF_GROUP_WGRAD_MINBLK=64" "RSSF_GROUP_WGRAD_TPB=32 RSSF_GROUP_WGRAD_MINBLK=128" "A=1"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
