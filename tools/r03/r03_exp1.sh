#!/bin/bash
# round-3 session 1: new parity tests + the strand / part / overlap matrix on the headline step
tag=r03a
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parrot.py tests/test_gpu_fullshape.py -q -m gpu --timeout 600 -x \
  -k "strands or T800 or decode_1000" 2>&1 | tail -25 | tee gpurun_out/$tag/tests.log
timeout 1500 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_STRANDS:1,PARROT_QPART:0 \
  parts=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  parts_ov=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:1 \
  parts_ov_pad=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_LDS_PAD:65536 \
  s2=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s2_q0=PARROT_STRANDS:2,PARROT_QPART:0,PARROT_DW_OVERLAP:0 \
  s2_full224=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:224 \
  s2_full64=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:64 \
  s2_ov=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1 \
  s2_ov_pad=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_LDS_PAD:65536 \
  s4=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s4_q8=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0,GPU_MAX_HW_QUEUES:8 \
  s4_ov=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:1,GPU_MAX_HW_QUEUES:8 \
  s4_ov_pad=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_LDS_PAD:65536,GPU_MAX_HW_QUEUES:8 \
  s2_q50=PARROT_STRANDS:2,PARROT_QPART:50,PARROT_DW_OVERLAP:1 \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-220
# kernel stats + trace of the 2-strand + overlap variant
PARROT_STRANDS=2 PARROT_QPART=100 PARROT_DW_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p1 -- \
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > gpurun_out/$tag/prof_s2.json 2> gpurun_out/$tag/prof_s2.err
find gpurun_out/$tag/p1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/s2_kernel_stats.csv \;
find gpurun_out/$tag/p1 -name "*kernel_trace.csv" -size -40M -exec cp {} gpurun_out/$tag/s2_kernel_trace.csv \;
rm -rf gpurun_out/$tag/p1
cut -c1-160 gpurun_out/$tag/s2_kernel_stats.csv | head -12
