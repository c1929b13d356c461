#!/bin/bash
tag=r03c
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tools/probe_bin/graph_host_probe 2>&1 | tee gpurun_out/$tag/graph_host_probe.txt
timeout 1200 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_STRANDS:1,PARROT_QPART:0 \
  s2=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s2_hi=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_STRAND_PRIO:-1 \
  s2_full224=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:224 \
  s4=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0 \
  s4_full224=PARROT_STRANDS:4,PARROT_QPART:100,PARROT_DW_OVERLAP:0,PARROT_SK_FULL:224 \
  s1_ov_p0=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0 \
  s1_ov_p0_pad=PARROT_STRANDS:1,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0,PARROT_DW_LDS_PAD:65536 \
  s2_ov_p0=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0 \
  s2_ov_p0_pad=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0,PARROT_DW_LDS_PAD:65536 \
  s2_ov_p0_224=PARROT_STRANDS:2,PARROT_QPART:100,PARROT_DW_OVERLAP:1,PARROT_DW_PRIORITY:0,PARROT_SK_FULL:224 \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-200
PARROT_STRANDS=2 PARROT_QPART=100 PARROT_DW_OVERLAP=1 PARROT_DW_PRIORITY=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p1 -- \
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-roofline --no-parity --no-secondary > gpurun_out/$tag/prof.json 2> gpurun_out/$tag/prof.err
find gpurun_out/$tag/p1 -name "*kernel_trace.csv" -size -40M -exec cp {} gpurun_out/$tag/s2ov_kernel_trace.csv \;
rm -rf gpurun_out/$tag/p1
