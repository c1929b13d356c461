#!/bin/bash
tag=r03u
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -s -k "layers_and_row_blocks or feedback_speaker or dataflow" 2>&1 | grep -i "sticky\|passed\|failed\|FAILED" | tee gpurun_out/$tag/persist_sites.log
timeout 600 python -m pytest tests/test_gpu_fullshape.py -q -m gpu --timeout 600 -s -k "decode_1000" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/$tag/decode1000.log
tail -16 gpurun_out/$tag/decode1000.log
