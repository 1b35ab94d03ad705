#!/bin/bash
# does the folded C = 48 attention backward (lesson 21) still mis-compile with this toolchain?  (variant: tools/ab_lib.sh foldall win_attn_bwd.hip -DRSSF_BWD_FOLD_ALL=1)
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in foldall ship; do
  lib=""; [ "$v" != "ship" ] && lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_$v.so
  echo "== $v" | tee -a $o/fold.txt
  RSSF_LIB_OVERRIDE=$lib timeout 900 python tools/attn_c48_dbg.py 2>&1 | grep "^0" | tee -a $o/fold.txt
  RSSF_LIB_OVERRIDE=$lib timeout 1200 python -m pytest tests/test_gpu_block.py tests/test_gpu_attention.py -q -k "48 or large or full_geometry" 2>&1 | tail -8 | tee -a $o/fold.txt
done
