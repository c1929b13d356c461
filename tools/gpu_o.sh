#!/bin/bash
mkdir -p gpurun_out/r02o
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --timeout 600 -x -k "wide or cfg4 or small" 2>&1 | tail -12 | tee gpurun_out/r02o/tests.log
for wk in 0 1; do
PARROT_WK=$wk timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02o/cfg4_wk$wk.json 2> gpurun_out/r02o/cfg4_wk$wk.err
python -c "
import json
d=json.load(open('gpurun_out/r02o/cfg4_wk$wk.json')); r=d['roofline']; print('wk$wk', d['value'], d['ms_per_step'], r['kernel_time_ms_per_step'], r['avg_launch_us'], r['frac'], d['final_cost'])"
done
