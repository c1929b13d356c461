#!/bin/bash
tag=r03q
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;5,6p' | tee gpurun_out/$tag/probe.log
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-secondary --no-dense --no-roofline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/$tag/bench_short.json
( time timeout 1700 python -m pytest tests -q -m gpu --timeout 600 -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/$tag/tests_full.log
