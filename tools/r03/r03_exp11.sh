#!/bin/bash
tag=r03k
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 120 -x -k "balanced_wavefront or (scan_schedules and gru)" 2>&1 | tail -8 | tee gpurun_out/$tag/tests.log
for cfg in "PARROT_SCHEDULE=5" "PARROT_SCHEDULE=6" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=1" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=4" "PARROT_SCHEDULE=6 PARROT_S6_ESPLIT=8" "PARROT_SCHEDULE=6 PARROT_S5_FULL=224"; do
  echo "== $cfg"; env $cfg timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;5,6p'
done | tee gpurun_out/$tag/probe.log
for s in 5 6; do
PARROT_SCHEDULE=$s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$s -- python tools/host_launch_probe.py > /tmp/prof$s.log 2>&1
f=$(find /tmp/prof$s -name "*kernel_stats.csv" | head -1)
echo "== schedule $s"; head -9 "$f" | cut -c1-180
cp "$f" gpurun_out/$tag/s${s}_kernel_stats.csv
done 2>&1 | tee gpurun_out/$tag/prof.log
