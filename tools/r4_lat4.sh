#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lattice.py -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -3 $o/pytest.txt
RSSF_LATTICE_WG8=0 timeout 600 python -m pytest tests/test_gpu_lattice.py -x -q > $o/pytest4.txt 2>&1; echo "pytest(wg8=0) rc $?" >> $o/pytest4.txt; tail -2 $o/pytest4.txt
for b in 7 14 16; do for w in 1 0; do echo "B=$b WG8=$w $(RSSF_LATTICE_WG8=$w timeout 100 python tools/lattice_bench.py $b 2>&1 | grep 'LATTICE=1')" | tee -a $o/wg8.txt; done; done
