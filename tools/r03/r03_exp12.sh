#!/bin/bash
tag=r03l
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 120 -x -k "balanced_wavefront" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.log
for cfg in "PARROT_SCHEDULE=5 PARROT_ATT_ESPLIT=1" "PARROT_SCHEDULE=5 PARROT_ATT_ESPLIT=1 PARROT_SKA_LDS_PAD=65536" "PARROT_SCHEDULE=5 PARROT_SKA_LDS_PAD=65536" \
  "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=1" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=2" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=1 PARROT_SKA_LDS_PAD=65536" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=1 PARROT_SKA_LDS_PAD=24576"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;6p'
done | tee gpurun_out/$tag/probe.log
PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python tools/host_launch_probe.py > /tmp/prof6.log 2>&1
f=$(find /tmp/prof6 -name "*kernel_stats.csv" | head -1)
echo "== schedule 6 esplit 1"; head -5 "$f" | cut -c1-180
cp "$f" gpurun_out/$tag/s6_es1_kernel_stats.csv
