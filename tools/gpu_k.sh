#!/bin/bash
mkdir -p gpurun_out/r02k
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in 2 3; do
PARROT_SCHEDULE=$s timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > gpurun_out/r02k/cfg4_bf16_s$s.json 2> gpurun_out/r02k/cfg4_bf16_s$s.err
done
PARROT_SCHEDULE=2 timeout 300 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > gpurun_out/r02k/cfg2_bf16_s2.json 2> gpurun_out/r02k/cfg2_bf16_s2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02k/prof -- python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > gpurun_out/r02k/prof.json 2> gpurun_out/r02k/prof.err
find gpurun_out/r02k/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02k/cfg4_bf16_kernel_stats.csv \;
rm -rf gpurun_out/r02k/prof
for f in gpurun_out/r02k/*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
head -25 gpurun_out/r02k/cfg4_bf16_kernel_stats.csv
