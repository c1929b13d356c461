#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
export TMPDIR=/tmp
{ timeout 300 python tools/pm_debug.py; PM_GRAPH=0 timeout 300 python tools/pm_debug.py; } 2>&1 | grep -v Warning | tee gpurun_out/r03w/pm_debug.log
