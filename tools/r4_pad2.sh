#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -4 $o/pytest.txt
timeout 600 python tools/aten_prof.py > $o/aten.txt 2>&1; grep -A30 "ATen kernels" $o/aten.txt | cut -c1-300
