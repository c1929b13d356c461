#!/bin/bash
mkdir -p gpurun_out/r02m
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_samplernn.py -q -m gpu --timeout 300 -x 2>&1 | tail -15 | tee gpurun_out/r02m/tests.log
timeout 600 python -m pytest tests/test_gpu_fullshape.py -q -m gpu --timeout 500 -x -k cfg5 2>&1 | tail -8 | tee gpurun_out/r02m/tests_cfg5.log
timeout 300 python tools/bench_extra.py > gpurun_out/r02m/secondary.json 2> gpurun_out/r02m/secondary.err; tail -c 1500 gpurun_out/r02m/secondary.json; tail -3 gpurun_out/r02m/secondary.err
