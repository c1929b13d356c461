#!/bin/bash
tag=r03n
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 120 -x -k "cost_and_grads or scan_schedules or balanced_wavefront or strands_and_parts" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.log
for cfg in "PARROT_BWD_SPLIT=0" "PARROT_BWD_SPLIT=1"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;5,6p'
done | tee gpurun_out/$tag/probe.log
echo "== skbench4 <2,1> sweep"; PARROT_SK_TILE=2,1 tools/probe_bin/skbench4 2>&1 | grep sweep | tee gpurun_out/$tag/skbench4_tile21.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -- python tools/host_launch_probe.py > /tmp/prof5.log 2>&1
f=$(find /tmp/prof5 -name "*kernel_stats.csv" | head -1)
echo "== default (S5 + bwd split)"; head -6 "$f" | cut -c1-180
cp "$f" gpurun_out/$tag/s5_bwdsplit_kernel_stats.csv
