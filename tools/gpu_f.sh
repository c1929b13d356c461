#!/bin/bash
mkdir -p gpurun_out/r02f
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > gpurun_out/r02f/all_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r02f/all_tests.log
tail -n 25 gpurun_out/r02f/all_tests.log
for s in 4 0; do
PARROT_SCHEDULE=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02f/bench_s$s.json 2> gpurun_out/r02f/bench_s$s.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02f/bench_s$s.json").read().strip().splitlines()[-1])
print("schedule $s", d["value"], d["ms_per_step"], d["roofline"])
PY
done
