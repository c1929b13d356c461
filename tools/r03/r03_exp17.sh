#!/bin/bash
tag=r03r
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/decode_drift.py 2>&1 | grep -v Warning | tail -8 | tee gpurun_out/$tag/decode_drift.txt
