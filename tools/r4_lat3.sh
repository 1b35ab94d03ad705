#!/bin/bash
# lattice kernel: parity, then timing ablations (RSSF_LATTICE_ABL: 1 no weight loads, 2 no A reads, 4 no staging, 8 no MFMAs, 7 = 1+2+4)
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lattice.py -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -3 $o/pytest.txt
for b in 14 16; do for abl in 0 1 2 4 8 7; do echo "B=$b ABL=$abl $(RSSF_LATTICE_ABL=$abl timeout 100 python tools/lattice_bench.py $b 2>&1 | grep 'LATTICE=1')" | tee -a $o/abl.txt; done; done
