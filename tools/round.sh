#!/bin/bash
# ONE harness for the measurements of a round (run on the GPU box through gpurun, from the repository root).  It replaces the
# r3_* / r4_* / r5_* one-off scripts of earlier rounds; every mode writes under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#
#   tools/round.sh check    <tag> "<pytest args>"                      a part of the GPU suite (default: the whole suite) + smoke
#   tools/round.sh ab       <tag> VARIANT [VARIANT ...]                bench.py (30 steps) under each variant, interleaved twice; a VARIANT is
#                                                                      "ship", "lib:<name>" (lib/ab/librssf_<name>.so, tools/ab_lib_flags.sh)
#                                                                      or "ENV=val[ ENV=val..]"
#   tools/round.sh census   <tag> [ENV=val ...]                        eager kernel trace of 2 + 3 steps -> kernel_stats.csv, census.txt, launches.txt
#   tools/round.sh timeline <tag> [ENV=val ...]                        kernel trace of the REPLAYED step -> timeline.txt (per-queue busy / idle / gaps)
#   tools/round.sh traffic  <tag>                                      fabric bytes per kernel: the attention pair, the kernels of traffic_cmd.py, one step
#   tools/round.sh dp       <tag>                                      forced 1-rank data-parallel step, both SyncBN exchanges (tools/dp_graph.py)
#   tools/round.sh large    <tag>                                      BASELINE config 4 line (Large 4 x 3 x 1024^2)
#   tools/round.sh final    <tag> [skip-tests]                         all of the above + the default bench record: the end-of-round record
mode=$1; tag=$2; shift 2
o=gpurun_out/$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT

check() {
  t0=$(date +%s); timeout 2400 python -m pytest ${1:-tests -m gpu} -x -q > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
  echo "pytest wall $(( $(date +%s) - t0 )) s" > $o/pytest_time.txt
  [ -z "$1" ] && timeout 600 python __graft_entry__.py smoke > $o/smoke.txt 2>&1
  tail -4 $o/pytest.txt; cat $o/pytest_time.txt; [ -f $o/smoke.txt ] && tail -2 $o/smoke.txt
}
ab() {
  for rep in 1 2; do
    for v in "$@"; do
      case "$v" in
        ship) e="RSSF_LIB_OVERRIDE=";;
        lib:*) e="RSSF_LIB_OVERRIDE=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_${v#lib:}.so";;
        *) e="$v";;
      esac
      r=$(env $e timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
      echo "$v  $r" | tee -a $o/ab.txt
    done
  done
}
census() {
  env "$@" RSSF_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - $o <<'PY'
import glob, sys, shutil
o = sys.argv[1]
for f in glob.glob(o + '/prof/**/r1_kernel_stats.csv', recursive=True): shutil.copy(f, o + '/kernel_stats.csv')
PY
  python tools/prof_step.py $tag/prof 90 > $o/census.txt 2>&1
  python tools/prof_launches.py $tag > $o/launches.txt 2>&1
  find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
  head -14 $o/census.txt
}
timeline() {
  mkdir -p $o/tl
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/tl -o r1 -- python bench.py --steps 4 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python tools/graph_timeline.py $tag/tl 40 > $o/timeline.txt 2>&1
  find $o/tl -name "*kernel_trace.csv" -delete; find $o/tl -name "*.db" -delete
  cut -c1-330 $o/timeline.txt | head -70
}
traffic() {
  tools/hbm_traffic.sh ${tag}_hbm_attn python tools/attn_bwd.py 3 > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_attn.txt $o/hbm_traffic_attn.txt
  tools/hbm_traffic.sh ${tag}_hbm_kernels python tools/traffic_cmd.py > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_kernels.txt $o/hbm_traffic_kernels.txt
  tools/hbm_traffic.sh ${tag}_hbm_step python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cp gpurun_out/${tag}_hbm_step.txt $o/hbm_traffic_step.txt
  python tools/wgrad_jobs.py > $o/wgrad_jobs.txt 2>&1
  head -5 $o/hbm_traffic_attn.txt
}
dp() {
  rm -f $o/dp_forced_1rank.txt
  for v in "A=1" "RSSF_SYNCBN=rccl"; do echo "== forced 1-rank data parallel, $v" >> $o/dp_forced_1rank.txt; env $v timeout 400 python tools/dp_graph.py 2>&1 | grep "ms/step\|SyncBN\|dp buckets" >> $o/dp_forced_1rank.txt; done
  cat $o/dp_forced_1rank.txt
}
large() { timeout 600 python bench.py --variant large --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_large.json 2>/dev/null; cut -c1-300 $o/bench_large.json; }

case $mode in
  check) check "$1";;
  ab) ab "$@";;
  census) census "$@";;
  timeline) timeline "$@";;
  traffic) traffic;;
  dp) dp;;
  large) large;;
  final)
    [ "$1" = "skip-tests" ] || check ""
    timeout 900 python bench.py > $o/bench.json 2> $o/bench.err; cut -c1-300 $o/bench.json
    census; timeline; traffic; dp; large;;
  *) echo "unknown mode $mode"; exit 2;;
esac
