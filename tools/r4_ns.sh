#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_model.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -4 $o/pytest.txt
for v in "RSSF_NEAREST_SUM=0" "RSSF_NEAREST_SUM=1" "RSSF_NEAREST_SUM=0" "RSSF_NEAREST_SUM=1"; do echo "== $v" >> $o/bench.txt; env $v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> $o/bench.txt; done
cat $o/bench.txt
