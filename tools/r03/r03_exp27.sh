#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03aa
export TMPDIR=/tmp
run() { env $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-dense --no-parity --no-secondary --no-roofline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'], d['value'], d['config']['scan_schedule'][:3], d['final_cost'])" | tee -a gpurun_out/r03aa/variants.log; }
run "PARROT_SCHEDULE=0" "--dtype bf16"
run "PARROT_SCHEDULE=5" "--dtype bf16"
run "PARROT_SCHEDULE=0" "--config cfg4 --dtype f32"
run "PARROT_SCHEDULE=5" "--config cfg4 --dtype f32"
run "PARROT_SCHEDULE=0" "--L 3"
run "PARROT_SCHEDULE=5" "--L 3"
