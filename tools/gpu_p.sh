#!/bin/bash
mkdir -p gpurun_out/r02p
cd $GRAFT_REPO_ROOT
for v in halfA halfB; do
PARROT_HIP_LIB=$GRAFT_REPO_ROOT/tmp_libs/libparrot_$v.so timeout 300 python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-dense > gpurun_out/r02p/cfg4_$v.json 2> gpurun_out/r02p/cfg4_$v.err
python -c "
import json
d=json.load(open('gpurun_out/r02p/cfg4_$v.json')); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['kernel_time_ms_per_step'], r['avg_launch_us'])"
done
