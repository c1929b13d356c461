#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ad
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bricks.py -q -m gpu --timeout 300 -k "gru or Gated or Bidirectional or encoder or brick" 2>&1 | tail -4 | tee gpurun_out/r03ad/tests.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python tools/host_launch_probe.py > /tmp/pr.log 2>&1
f=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
grep -i "rg_\|Name" "$f" | cut -c1-160 | tee gpurun_out/r03ad/rowgru_stats.txt
for m in 1 0; do
PARROT_GRU_ROWWISE=$m timeout 300 python bench.py --no-cpu-baseline --no-parity --no-secondary --no-dense --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rowwise $m', d['ms_per_step'], d['value'], d['final_cost'])" | tee -a gpurun_out/r03ad/bench.log
done
