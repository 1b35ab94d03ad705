#!/bin/bash
# forced 1-rank data-parallel step (RCCL communicators, buckets, SyncBN exchanges, hipGraph) under the stream / plan switches
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; echo "== $tag" >> $o/dp.txt; env "$@" timeout 400 python tools/dp_graph.py 2>&1 | grep "ms/step\|losses\|SyncBN\|dp buckets\|rror\|Traceback" >> $o/dp.txt; }
run p2p A=1
run rccl RSSF_SYNCBN=rccl
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_trainer.py -x -q > $o/pytest_dp.txt 2>&1; echo "rc $?" >> $o/pytest_dp.txt
cat $o/dp.txt; tail -5 $o/pytest_dp.txt
