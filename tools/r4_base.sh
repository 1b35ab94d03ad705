#!/bin/bash
# round-4 record of the current state: default bench, eager kernel stats + census, fabric traffic of the attention kernels
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - $o <<'PY'
import glob, sys, shutil
o = sys.argv[1]
for f in glob.glob(o + '/prof/**/r1_kernel_stats.csv', recursive=True): shutil.copy(f, o + '/kernel_stats.csv')
PY
tag=$(basename $o)
python tools/prof_step.py $tag/prof 90 > $o/census.txt 2>&1
tools/hbm_traffic.sh ${tag}_hbm_attn python tools/attn_bwd.py 3 > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_attn.txt $o/hbm_traffic_attn.txt
if [ -n "$2" ]; then timeout 1500 python -m pytest tests -m gpu -x -q $2 > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -3 $o/pytest.txt; fi
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
cut -c1-400 $o/bench.json; head -60 $o/census.txt
