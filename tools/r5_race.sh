#!/bin/bash
# does the round-2 gate anomaly (lesson 23) still reproduce with an SLP-vectorised gate.o on this box?
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in gateslp ship; do
  lib=""; [ "$v" != "ship" ] && lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_$v.so
  echo "== $v: gate_race" | tee -a $o/race.txt
  RSSF_LIB_OVERRIDE=$lib timeout 600 python tools/gate_race.py 400 2>&1 | tail -6 | tee -a $o/race.txt
  echo "== $v: replay_race" | tee -a $o/race.txt
  RSSF_LIB_OVERRIDE=$lib timeout 900 python tools/replay_race.py 80 2 128 2>&1 | tail -12 | tee -a $o/race.txt
done
