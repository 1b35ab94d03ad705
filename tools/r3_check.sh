#!/bin/bash
# usage: tools/r3_check.sh <tag> [pytest args]   GPU test suite + default bench (no CPU baseline) + ATen residue listing
tag=$1; shift
o=gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q "$@" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err
timeout 300 python tools/aten_prof.py > $o/aten.txt 2>&1
tail -5 $o/pytest.txt; cut -c1-400 $o/bench.json
