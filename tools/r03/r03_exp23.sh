#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03x
export TMPDIR=/tmp
PARROT_PM_DUMP=1 timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -s -x -k "layers_and_row_blocks" 2>&1 | grep -i "\[pm\]\|sticky\|xcc\|nwg\|passed\|failed\|FAILED" | tee gpurun_out/r03x/persist_dump2.log
