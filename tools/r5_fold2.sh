#!/bin/bash
o=gpurun_out/$1; mkdir -p $o; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_$v.so
  echo "== $v $(RSSF_LIB_OVERRIDE=$lib timeout 300 python tools/attn_c48_dbg.py 2>&1 | grep -E '^0 (gy|attn.v_proj.weight|attn.q_proj.weight|weight_levels.weight)' | awk '{printf "%s %s |f32| %s ; ", $2, $3, $5}')" | tee -a $o/fold2.txt
done
