#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lattice.py -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -3 $o/pytest.txt
for b in 7 16; do echo "B=$b $(RSSF_LATTICE=1 timeout 100 python tools/lattice_bench.py $b 2>&1 | grep 'LATTICE=1')" | tee -a $o/bench.txt; done
for v in 1 0; do RSSF_LATTICE=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/RSSF_LATTICE=$v /" | tee -a $o/ab.txt; done
