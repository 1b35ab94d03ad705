#!/bin/bash
# multi-tile halo kernels (RSSF_HALO_NTB): parity under the switch, stand-alone group bench, step A/B
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RSSF_HALO_NTB=3 timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_conv.py -m gpu -x -q > $o/pytest_ntb3.txt 2>&1; echo "rc $?" >> $o/pytest_ntb3.txt
tail -5 $o/pytest_ntb3.txt
for n in 0 1 2 4 8; do echo "== RSSF_HALO_NTB=$n" >> $o/group_bench.txt; RSSF_HALO_NTB=$n GB_ONLY=3x3 timeout 300 python tools/group_bench.py 4 30 >> $o/group_bench.txt 2>&1; done
cat $o/group_bench.txt
for n in 0 2 4 0 2; do echo "== RSSF_HALO_NTB=$n" >> $o/bench_ab.txt; RSSF_HALO_NTB=$n timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench_ab.txt; done
cat $o/bench_ab.txt
