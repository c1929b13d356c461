#!/bin/bash
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_persist.py -q -m gpu -x --timeout 300 > gpurun_out/r02b/persist_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r02b/persist_tests.log
tail -n 40 gpurun_out/r02b/persist_tests.log
