#!/bin/bash
tag=r03v
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -i "compute unit\|gfx950\|Marketing" | head -6
PARROT_PM_DUMP=1 timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -s -k "layers_and_row_blocks" 2>&1 | grep -i "sticky\|xcc\|nwg\|passed\|failed\|FAILED" | tee gpurun_out/$tag/persist_dump.log
