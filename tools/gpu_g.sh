#!/bin/bash
mkdir -p gpurun_out/r02g
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_persist.py -q -m gpu -x --timeout 300 2>&1 | tail -3 | tee gpurun_out/r02g/persist_tests.log
MODE=decode timeout 300 python tools/pm_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02g/pm_timing_decode.log
MODE=train timeout 300 python tools/pm_timing.py 2>&1 | grep -v amdgpu.ids | grep "forward\|per tick" | tee gpurun_out/r02g/pm_timing_train.log
