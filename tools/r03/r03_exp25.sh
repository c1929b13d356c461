#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parrot.py tests/test_gpu_bf16.py -q -m gpu --timeout 300 -k "balanced_wavefront or scan_schedules or bf16 or lstm" 2>&1 | tail -12 | tee gpurun_out/r03z/tests.log
for s in 0 5; do
PARROT_SCHEDULE=$s timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-dense --no-parity --no-secondary --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 schedule $s', d['ms_per_step'], d['value'], d['config']['scan_schedule'][:40], d['roofline']['frac'], d['final_cost'])" | tee -a gpurun_out/r03z/cfg4.log
done
