#!/bin/bash
tag=r03g
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
for v in base af af3 af4; do echo "== $v"; tools/probe_bin/skbench3_$v; tools/probe_bin/skbench3_$v 32; done 2>&1 | tee gpurun_out/$tag/skbench3.txt
