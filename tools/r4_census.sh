#!/bin/bash
# eager kernel census of one step (rocprofv3 kernel trace) under the environment given as extra arguments: tools/r4_census.sh <tag> [VAR=val ...]
tag=$1; shift
o=gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
env "$@" RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - $o <<'PY'
import glob, sys, shutil
o = sys.argv[1]
for f in glob.glob(o + '/prof/**/r1_kernel_stats.csv', recursive=True): shutil.copy(f, o + '/kernel_stats.csv')
PY
python tools/prof_step.py $tag/prof 90 > $o/census.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
head -12 $o/census.txt
