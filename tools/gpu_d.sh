#!/bin/bash
mkdir -p gpurun_out/r02d
cd $GRAFT_REPO_ROOT
timeout 300 python tools/pm_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/pm_timing.log
