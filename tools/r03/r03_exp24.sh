#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 2>&1 | tail -15 | tee gpurun_out/r03y/persist2.log
