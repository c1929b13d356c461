#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ab
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bricks.py -q -m gpu --timeout 300 -k "gru or Gated or Bidirectional or encoder or brick" 2>&1 | tail -6 | tee gpurun_out/r03ab/tests.log
timeout 600 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 300 -k "cost_and_grads_layers or feedback_speaker_ragged or nonliteral" 2>&1 | tail -3 | tee -a gpurun_out/r03ab/tests.log
for m in 1 0; do
PARROT_GRU_ROWWISE=$m timeout 300 python bench.py --no-cpu-baseline --no-parity --no-secondary --no-dense --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rowwise $m', d['ms_per_step'], d['value'], d['final_cost'])" | tee -a gpurun_out/r03ab/bench.log
done
