#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
lib=$GRAFT_REPO_ROOT/representationlearning_amd/lib/ab/librssf_gateslp.so
for i in 1 2 3; do
  RSSF_LIB_OVERRIDE=$lib timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -k "repeats_under_its_own_concurrency" 2>&1 | tail -3 | tee -a $o/race2.txt
done
echo "== branch-by-branch walk, three side streams (the round-2 stream structure)" | tee -a $o/race2.txt
RSSF_LOCKSTEP=0 RSSF_LIB_OVERRIDE=$lib timeout 900 python tools/replay_race.py 150 2 128 2>&1 | tail -4 | tee -a $o/race2.txt
RSSF_LOCKSTEP=0 RSSF_FORK_FUSE=0 RSSF_LIB_OVERRIDE=$lib timeout 900 python tools/replay_race.py 150 2 128 2>&1 | tail -4 | tee -a $o/race2.txt
python - <<'PY' | tee -a $o/race2.txt
import torch, subprocess
print("torch", torch.__version__, "hip", torch.version.hip)
print(subprocess.run("cat /sys/module/amdgpu/version 2>/dev/null; uname -r; rocm-smi --showfwinfo 2>/dev/null | grep -i -E 'mec|sdma|smc|rlc' | head -8", shell=True, capture_output=True, text=True).stdout)
PY
