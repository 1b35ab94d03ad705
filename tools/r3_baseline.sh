#!/bin/bash
# round-3 baseline pass on the GPU box: config-4 / config-5 numbers, ATen residue, replay timeline, eager census
o=gpurun_out/r3a; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --variant large --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_large.json 2> $o/bench_large.err
timeout 300 python tools/cam_bench.py > $o/cam.json 2> $o/cam.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/cam_prof -o r1 -- python tools/cam_bench.py 32 > /dev/null 2>&1
timeout 300 python tools/aten_prof.py > $o/aten.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a_graph -o r1 -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline > $o/graph_bench.json 2>/dev/null
python tools/graph_timeline.py r3a_graph > $o/timeline.txt 2>&1
tools/prof.sh r3a_prof > $o/prof_eager.txt 2>&1
python tools/prof_step.py r3a_prof 90 > $o/census.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
ls -la $o
