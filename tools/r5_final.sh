#!/bin/bash
# end-of-round record: GPU test suite, smoke, default bench, eager kernel stats + census, fabric traffic, forced 1-rank DP, config 4 line
#   tools/r5_final.sh <tag> [skip-tests]
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -z "$2" ] || [ "$2" = "tests-only" ]; then
  t0=$(date +%s); timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
  echo "pytest wall $(( $(date +%s) - t0 )) s" > $o/pytest_time.txt
  timeout 600 python __graft_entry__.py smoke > $o/smoke.txt 2>&1
fi
if [ "$2" = "tests-only" ]; then tail -3 $o/pytest.txt; cat $o/pytest_time.txt; tail -2 $o/smoke.txt; exit 0; fi
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - $o <<'PY'
import glob, sys, shutil
o = sys.argv[1]
for f in glob.glob(o + '/prof/**/r1_kernel_stats.csv', recursive=True): shutil.copy(f, o + '/kernel_stats.csv')
PY
tag=$(basename $o)
python tools/prof_step.py $tag/prof 90 > $o/census.txt 2>&1
python tools/prof_launches.py $tag > $o/launches.txt 2>&1
tools/hbm_traffic.sh ${tag}_hbm_attn python tools/attn_bwd.py 3 > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_attn.txt $o/hbm_traffic_attn.txt
tools/hbm_traffic.sh ${tag}_hbm_kernels python tools/traffic_cmd.py > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_kernels.txt $o/hbm_traffic_kernels.txt
tools/hbm_traffic.sh ${tag}_hbm_step python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_step.txt $o/hbm_traffic_step.txt
python tools/wgrad_jobs.py > $o/wgrad_jobs.txt 2>&1
for v in "A=1" "RSSF_SYNCBN=rccl"; do echo "== forced 1-rank data parallel, $v" >> $o/dp_forced_1rank.txt; env $v timeout 400 python tools/dp_graph.py 2>&1 | grep "ms/step\|SyncBN\|dp buckets" >> $o/dp_forced_1rank.txt; done
timeout 600 python bench.py --variant large --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_large.json 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
tail -3 $o/pytest.txt; cat $o/pytest_time.txt; tail -2 $o/smoke.txt; cut -c1-300 $o/bench.json; cat $o/dp_forced_1rank.txt; head -8 $o/census.txt
