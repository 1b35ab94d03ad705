#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -x -q -k "multi_tile" > $o/pytest_mt.txt 2>&1; tail -3 $o/pytest_mt.txt
timeout 600 python -m pytest tests/test_gpu_scd.py -m gpu -x -q -k "graphed" > $o/pytest_scd.txt 2>&1; tail -3 $o/pytest_scd.txt
timeout 600 python tools/aten_prof.py > $o/aten.txt 2>&1; tail -40 $o/aten.txt | cut -c1-330
