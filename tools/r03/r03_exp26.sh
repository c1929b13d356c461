#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parrot.py tests/test_gpu_bf16.py -q -m gpu --timeout 300 -k "balanced_wavefront or (scan_schedules and lstm) or cfg4 or bf16_small" 2>&1 | tail -3 | tee gpurun_out/r03z/tests2.log
for cfg in "PARROT_SCHEDULE=5" "PARROT_SCHEDULE=5 PARROT_SKA_ATT_LAST=0" "PARROT_SCHEDULE=5 PARROT_S5_SPLIT=0"; do
env $cfg timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-dense --no-parity --no-secondary --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 $cfg', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['final_cost'])" | tee -a gpurun_out/r03z/cfg4.log
done
PARROT_SCHEDULE=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -- python bench.py --config cfg4 --no-cpu-baseline --no-dense --no-parity --no-secondary --no-roofline --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-150; cp "$f" gpurun_out/r03z/cfg4_s5_kernel_stats.csv
