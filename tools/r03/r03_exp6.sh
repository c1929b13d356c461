#!/bin/bash
tag=r03f
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in base d4 p4 p6; do echo "== $v"; tools/probe_bin/skbench3_$v; done 2>&1 | tee gpurun_out/$tag/skbench3.txt
timeout 900 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_STRANDS:1,PARROT_QPART:0 \
  p4=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_p4.so \
  p6=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_p6.so \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-200
PARROT_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_p4.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullshape.py -q -m gpu --timeout 600 -x -k "not T800 and not decode_1000 and not cfg5" 2>&1 | tail -5 | tee gpurun_out/$tag/tests_p4.log
