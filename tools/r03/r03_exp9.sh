#!/bin/bash
# schedule 5 (balanced wavefront, attention + input projections in one heterogeneous launch): parity, then timing
tag=r03i
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 300 -x -k "scan_schedules and gru" 2>&1 | tail -5 | tee gpurun_out/$tag/tests.log
timeout 900 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_SCHEDULE:0 \
  s5=PARROT_SCHEDULE:5 \
  s5_last=PARROT_SCHEDULE:5,PARROT_SKA_ATT_LAST:1 \
  s5_full224=PARROT_SCHEDULE:5,PARROT_S5_FULL:224 \
  s5_full224_last=PARROT_SCHEDULE:5,PARROT_S5_FULL:224,PARROT_SKA_ATT_LAST:1 \
  s5_es4=PARROT_SCHEDULE:5,PARROT_ATT_ESPLIT:4 \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-220
PARROT_SCHEDULE=5 timeout 300 python tools/host_launch_probe.py 2>&1 | tail -8 | tee gpurun_out/$tag/probe.log
