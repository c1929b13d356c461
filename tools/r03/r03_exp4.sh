#!/bin/bash
tag=r03d
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=tools/probe_bin/graph_host_probe
( echo "== default"; $P
  echo "== 64-block kernels (a quarter of the chip)"; $P 600 1000 64
  echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $P
  echo "== DEBUG_HIP_FORCE_GRAPH_QUEUES=4"; DEBUG_HIP_FORCE_GRAPH_QUEUES=4 $P
  echo "== DEBUG_HIP_DYNAMIC_QUEUES=0"; DEBUG_HIP_DYNAMIC_QUEUES=0 $P
  echo "== DEBUG_HIP_DYNAMIC_QUEUES=1"; DEBUG_HIP_DYNAMIC_QUEUES=1 $P
  echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 $P
  echo "== AMD_DIRECT_DISPATCH=0"; AMD_DIRECT_DISPATCH=0 $P
  echo "== HSA_ENABLE_SDMA... ROC_SYSTEM_SCOPE_SIGNAL=0"; ROC_SYSTEM_SCOPE_SIGNAL=0 $P
) 2>&1 | tee gpurun_out/$tag/graph_probe.txt
# step-kernel pipelining depth variants on the headline step
timeout 900 python tools/exp_matrix.py gpurun_out/$tag/matrix.json \
  base=PARROT_STRANDS:1,PARROT_QPART:0 \
  d3=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_d3.so \
  d4=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_d4.so \
  d6=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_d6.so \
  blk=PARROT_STRANDS:1,PARROT_QPART:0,PARROT_HIP_LIB:$GRAFT_REPO_ROOT/tools/probe_bin/libparrot_blk.so \
  2>&1 | tee gpurun_out/$tag/matrix.log | cut -c1-200
