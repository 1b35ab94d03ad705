#!/bin/bash
# in-step A/B of one environment switch of the Python side:  tools/r5_env_ab.sh <outdir> "<pytest args>" VAR
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then
  timeout 1500 python -m pytest $2 -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -15 $o/pytest.txt
fi
for rep in 1 2; do
  for v in 1 0; do
    env $3=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>$o/bench_err_$v.txt | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$3=$v /" | tee -a $o/ab.txt
  done
done
tail -3 $o/bench_err_1.txt
