#!/bin/bash
# round 5: the transposed-copy weight gradient of the MLP tap sum - parity, then in-step A/B (RSSF_WGRAD_PLANES=1 / 0)
#   tools/r5_planes.sh <outdir> "<pytest args>"
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then
  timeout 1500 python -m pytest $2 -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -15 $o/pytest.txt
fi
for rep in 1 2; do
  for v in 1 0; do
    RSSF_WGRAD_PLANES=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>$o/bench_err_$v.txt | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/planes=$v /" | tee -a $o/ab.txt
  done
done
tail -3 $o/bench_err_1.txt
