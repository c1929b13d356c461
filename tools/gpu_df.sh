#!/bin/bash
mkdir -p gpurun_out/df
cd $GRAFT_REPO_ROOT
PARROT_PM_DATAFLOW=1 timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu --timeout 300 -x 2>&1 | tail -6 | tee gpurun_out/df/tests.log
for df in 0 1; do
PARROT_PM_DATAFLOW=$df PARROT_SCHEDULE=4 timeout 300 python bench.py --no-cpu-baseline --no-dense --no-roofline --steps 5 --warmup 2 > gpurun_out/df/b$df.json 2> gpurun_out/df/b$df.err
python -c "
import json
d=json.load(open('gpurun_out/df/b$df.json')); print('sched4 dataflow=$df', d['value'], d['ms_per_step'], d['final_cost'])"
PARROT_PM_DATAFLOW=$df timeout 300 python tools/bench_extra.py > gpurun_out/df/s$df.json 2> gpurun_out/df/s$df.err; python -c "
import json
d=json.load(open('gpurun_out/df/s$df.json')); print('decode dataflow=$df', d['decode_cfg3']['us_per_step'])"
done
