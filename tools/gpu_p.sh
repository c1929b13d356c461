#!/bin/bash
mkdir -p gpurun_out/r02p
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02p/cfg4_v4.json 2> gpurun_out/r02p/cfg4_v4.err
python -c "
import json
d=json.load(open('gpurun_out/r02p/cfg4_v4.json')); r=d['roofline']; print('v4', d['value'], d['ms_per_step'], r['kernel_time_ms_per_step'], r['avg_launch_us'], r['frac'], d['final_cost'])"
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu --timeout 600 -x -k "wide or cfg4" 2>&1 | tail -3
