#!/bin/bash
tag=r03p
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parrot.py tests/test_gpu_kernels.py -q -m gpu --timeout 120 -x -k "cost_and_grads or attention or scan_schedules" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.log
timeout 120 python tools/host_launch_probe.py 2>&1 | grep "device is done\|rror" | sed -n '2,3p;5,6p' | tee gpurun_out/$tag/probe.log
{
echo "== base"; tools/probe_bin/skbench4_base
echo "== A operand as fragment-major 1 KB blocks, no ds_bpermute (timing probe, values wrong)"; tools/probe_bin/skbench4_af
echo "== base, tile <2,1>"; PARROT_SK_TILE=2,1 tools/probe_bin/skbench4_base | grep sweep
echo "== af, tile <2,1>"; PARROT_SK_TILE=2,1 tools/probe_bin/skbench4_af | grep sweep
} 2>&1 | tee gpurun_out/$tag/skbench4_af.txt
