#!/bin/bash
# kernel census of one eager Large step (config 4: 4 x 3 x 1024 x 1024): tools/census_large.sh <tag>
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RSSF_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --variant large --size 1024 --batch 4 --steps 3 --warmup 2 --no-cpu-baseline > $o/bench_eager.json 2>$o/err.txt
python tools/prof_step.py $1/prof 70 > $o/census.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
head -75 $o/census.txt
