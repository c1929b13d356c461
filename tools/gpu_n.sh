#!/bin/bash
mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02n/prof -- python tools/bench_extra.py > gpurun_out/r02n/secondary.json 2> gpurun_out/r02n/err.txt
find gpurun_out/r02n/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02n/secondary_kernel_stats.csv \;
rm -rf gpurun_out/r02n/prof
head -14 gpurun_out/r02n/secondary_kernel_stats.csv | cut -c1-200
