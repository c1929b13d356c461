#!/bin/bash
# GPU call A (round 2): persistent-phase probe + the new parity tests + the whole GPU suite.
mkdir -p gpurun_out/r02a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
( timeout 180 tools/bin/persist_probe 2400 256 > gpurun_out/r02a/probe.log 2>&1; echo "rc=$?" >> gpurun_out/r02a/probe.log ) 
( timeout 120 tools/bin/persist_probe 2400 512 > gpurun_out/r02a/probe512.log 2>&1; echo "rc=$?" >> gpurun_out/r02a/probe512.log )
timeout 1500 python -m pytest tests/test_gpu_fullshape.py tests/test_gpu_ref_golden.py -q -m gpu -x --timeout 900 > gpurun_out/r02a/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r02a/new_tests.log
timeout 900 python -m pytest tests -q -m gpu --timeout 600 --deselect tests/test_gpu_fullshape.py --deselect tests/test_gpu_ref_golden.py > gpurun_out/r02a/all_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r02a/all_tests.log
tail -5 gpurun_out/r02a/probe.log gpurun_out/r02a/new_tests.log gpurun_out/r02a/all_tests.log
