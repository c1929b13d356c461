#!/bin/bash
# One GPU-box session: the -m gpu suite, smoke(), the headline bench, the cfg4 (bf16) line and the secondary benches.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_check.sh r02z'
tag=${1:-check}
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -15 | tee gpurun_out/$tag/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/$tag/smoke.log
timeout 600 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; tail -c 400 gpurun_out/$tag/bench.json
timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-dense > gpurun_out/$tag/bench_cfg4.json 2> gpurun_out/$tag/bench_cfg4.err
timeout 300 python tools/bench_extra.py > gpurun_out/$tag/secondary.json 2> gpurun_out/$tag/secondary.err; cat gpurun_out/$tag/secondary.json
python - <<PY
import json
for n in ("bench", "bench_cfg4"):
    try:
        d = json.load(open("gpurun_out/$tag/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
