#!/bin/bash
mkdir -p gpurun_out/r02c
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in 0 4; do
PARROT_SCHEDULE=$s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/prof$s -o s$s -- python bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/r02c/prof$s.log 2>&1
f=$(find gpurun_out/r02c/prof$s -name "*kernel_stats.csv" | head -1); echo "== $f"; head -14 "$f" | cut -c1-200
find gpurun_out/r02c/prof$s -name "*kernel_trace.csv" -delete
done
