#!/bin/bash
tag=r03t
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullshape.py -q -m gpu --timeout 600 -k "decode_1000" 2>&1 | tail -40 > gpurun_out/$tag/decode1000_head.log
tail -30 gpurun_out/$tag/decode1000_head.log
cd tmp_prev
timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -k "layers_and_row_blocks or feedback_speaker or dataflow" 2>&1 | tail -12 | tee ../gpurun_out/$tag/persist_prev_commit.log
cd ..
timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -x -k "layers_and_row_blocks" 2>&1 | tail -60 > gpurun_out/$tag/persist_head.log
PARROT_BWD_SPLIT=0 timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -k "layers_and_row_blocks" 2>&1 | tail -5 | tee gpurun_out/$tag/persist_head_nosplit.log
