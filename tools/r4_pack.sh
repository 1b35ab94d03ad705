#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trainer.py -x -q -k "pack or graph_replay_matches or direct_gradients or fused_mlp" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -3 $o/pytest.txt
timeout 600 python __graft_entry__.py smoke > $o/smoke.txt 2>&1; tail -7 $o/smoke.txt | cut -c1-400
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tee -a $o/ab.txt
timeout 300 python tools/aten_prof.py > $o/aten.txt 2>&1; tail -40 $o/aten.txt | cut -c1-220
