#!/bin/bash
# round-3 session 2: host cost of graph replays, threaded strand enqueue
tag=r03b
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 0 0" "1 100 0" "2 100 0" "2 100 1" "4 100 1" "2 0 1"; do
  set -- $cfg
  echo "== strands $1 qpart $2 threads $3" | tee -a gpurun_out/$tag/host.log
  PARROT_STRANDS=$1 PARROT_QPART=$2 PARROT_STRAND_THREADS=$3 PARROT_DW_OVERLAP=0 timeout 200 python tools/host_launch_probe.py 2>&1 | tail -6 | tee -a gpurun_out/$tag/host.log
done
timeout 900 python -m pytest tests/test_gpu_parrot.py -q -m gpu --timeout 600 -x -k "strands" 2>&1 | tail -8 | tee gpurun_out/$tag/tests.log
timeout 1200 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_STRANDS:1,PARROT_QPART:0 \
  s2t=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s2t_q0=PARROT_STRANDS:2,PARROT_QPART:0,PARROT_DW_OVERLAP:0 \
  s2t_full224=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:224 \
  s2t_nothreads=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_STRAND_THREADS:0 \
  s4t=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s4t_q0=PARROT_STRANDS:4,PARROT_QPART:0,PARROT_DW_OVERLAP:0 \
  s4t_full224=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:224 \
  s2t_ov=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1 \
  s1_ov_prio0=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0 \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-200
# trace of the 4-strand run (what do >= 3 active streams do?)
PARROT_STRANDS=4 PARROT_QPART=100 PARROT_DW_OVERLAP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p1 -- \
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-roofline --no-parity --no-secondary > gpurun_out/$tag/prof_s4.json 2> gpurun_out/$tag/prof_s4.err
find gpurun_out/$tag/p1 -name "*kernel_trace.csv" -size -40M -exec cp {} gpurun_out/$tag/s4_kernel_trace.csv \;
rm -rf gpurun_out/$tag/p1
