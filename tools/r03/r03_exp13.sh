#!/bin/bash
tag=r03m
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 120 -x -k "balanced_wavefront" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.log
for cfg in "PARROT_SCHEDULE=5" "PARROT_SCHEDULE=6" "PARROT_SCHEDULE=6 PARROT_S6_BTILE=21" "PARROT_SCHEDULE=6 PARROT_S6_BTILE=22" "PARROT_SCHEDULE=6 PARROT_S6_IB=0"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;6p'
done | tee gpurun_out/$tag/probe.log
PARROT_SCHEDULE=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python tools/host_launch_probe.py > /tmp/prof6.log 2>&1
f=$(find /tmp/prof6 -name "*kernel_stats.csv" | head -1)
echo "== schedule 6 default"; head -5 "$f" | cut -c1-180
cp "$f" gpurun_out/$tag/s6_kernel_stats.csv
