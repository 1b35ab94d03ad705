#!/bin/bash
# round 5: parity of the specialised convolution kernels + in-step A/B of library variants (tools/ab_lib.sh)
#   tools/r5_check.sh <outdir> "<pytest args>" [variant names...]
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then
  timeout 1500 python -m pytest $2 -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -5 $o/pytest.txt
fi
shift 2
for rep in 1 2; do
  for v in "$@"; do
    lib=""; [ "$v" != "ship" ] && lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_$v.so
    RSSF_LIB_OVERRIDE=$lib timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$v /" | tee -a $o/ab.txt
  done
done
