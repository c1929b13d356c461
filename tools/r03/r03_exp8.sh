#!/bin/bash
# Where do the ~6-8 us a step launch costs beyond its MFMA time go?  kernarg placement, descriptor through a device
# pointer, workgroups per CU (LDS pad), resident vs rotated operand sets, K sweep.
tag=r03h
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
b=tools/probe_bin/skbench4
{
echo "== base"; $b
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 $b
echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 $b
echo "== descriptor via device pointer"; SKB_PTR=1 $b
echo "== one workgroup per CU (LDS pad 96 KB)"; PARROT_SK_LDS_PAD=98304 $b
echo "== one operand set (cache-resident)"; SKB_NSETS=1 $b
echo "== one operand set + device pointer"; SKB_NSETS=1 SKB_PTR=1 $b
echo "== B=32"; $b 32
} 2>&1 | tee gpurun_out/$tag/skbench4.txt
