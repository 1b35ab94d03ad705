#!/bin/bash
# lattice kernel: parity tests, stand-alone timing (events + rocprof kernel durations), smoke with canaries, step bench A/B
o=gpurun_out/$1; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_conv.py -x -q -k "lattice or fused_mlp" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt; tail -15 $o/pytest.txt
timeout 300 python tools/lattice_bench.py > $o/lattice_bench.txt 2>&1; cat $o/lattice_bench.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o lb -- python tools/lattice_bench.py > /dev/null 2>&1
python - $o <<'PY'
import glob, sys, csv
for f in glob.glob(sys.argv[1] + '/prof/**/lb_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
timeout 600 python __graft_entry__.py smoke > $o/smoke.txt 2>&1; tail -4 $o/smoke.txt
for v in 1 0; do RSSF_LATTICE=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/RSSF_LATTICE=$v /" | tee -a $o/ab.txt; done
find gpurun_out -name "*kernel_trace.csv" -size +10M -delete; find gpurun_out -name "*.db" -size +10M -delete
