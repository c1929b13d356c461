#!/bin/bash
mkdir -p gpurun_out/r02l
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_ops.py -q -m gpu --timeout 600 -x -k "gemm" 2>&1 | tail -4 | tee gpurun_out/r02l/tests.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02l/$name.json 2> gpurun_out/r02l/$name.err; python -c "
import json
d=json.load(open('gpurun_out/r02l/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['kernel_time_ms_per_step'], r['avg_launch_us'])"; }
run base A=1
run t42 PARROT_SK_TILE=4,2
run t41 PARROT_SK_TILE=4,1
run t21 PARROT_SK_TILE=2,1
run t32 PARROT_SK_TILE=3,2
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > gpurun_out/r02l/cfg2_f32.json 2>gpurun_out/r02l/cfg2_f32.err; python -c "
import json
d=json.load(open('gpurun_out/r02l/cfg2_f32.json')); print('cfg2 f32', d['value'], d['ms_per_step'])"
