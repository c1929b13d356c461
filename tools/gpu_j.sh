#!/bin/bash
mkdir -p gpurun_out/r02j
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --timeout 600 -x 2>&1 | tail -25 | tee gpurun_out/r02j/tests.log
timeout 300 python bench.py --config cfg4 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02j/cfg4_f32.json 2> gpurun_out/r02j/cfg4_f32.err
timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02j/cfg4_bf16.json 2> gpurun_out/r02j/cfg4_bf16.err
timeout 300 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02j/cfg2_bf16.json 2> gpurun_out/r02j/cfg2_bf16.err
tail -c 600 gpurun_out/r02j/*.json; tail -3 gpurun_out/r02j/*.err
