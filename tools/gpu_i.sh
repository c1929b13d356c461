#!/bin/bash
mkdir -p gpurun_out/r02i
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_samplernn.py tests/test_gpu_persist.py -q -m gpu --timeout 300 2>&1 | tail -15 | tee gpurun_out/r02i/tests.log
