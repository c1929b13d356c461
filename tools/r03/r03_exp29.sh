#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ac
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python tools/host_launch_probe.py > /tmp/pr.log 2>&1
f=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
grep -i "rg_\|colsum\|Name" "$f" | cut -c1-160 | tee gpurun_out/r03ac/rowgru_stats.txt
cp "$f" gpurun_out/r03ac/probe_kernel_stats.csv
