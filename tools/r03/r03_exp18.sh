#!/bin/bash
tag=r03s
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests/test_gpu_fullshape.py tests/test_gpu_kernels.py tests/test_gpu_parrot.py tests/test_gpu_persist.py tests/test_gpu_ref_golden.py tests/test_gpu_samplernn.py -q -m gpu --timeout 600 -k "not cfg2_width and not cfg4_width and not T800" 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/$tag/tests_rest.log
