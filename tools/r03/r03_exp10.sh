#!/bin/bash
tag=r03j
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "PARROT_SCHEDULE=0" "PARROT_SCHEDULE=5" "PARROT_SCHEDULE=5 PARROT_SKA_ATT_LAST=1" "PARROT_SCHEDULE=5 PARROT_SKA_ATT_LAST=1 PARROT_S5_FULL=224"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/host_launch_probe.py 2>&1 | grep "device is done" | sed -n '2,3p;5,6p'
done | tee gpurun_out/$tag/probe.log
cd /tmp
for v in 0 1; do
PARROT_SCHEDULE=5 PARROT_SKA_ATT_LAST=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof$v -o s5 -- python $GRAFT_REPO_ROOT/tools/host_launch_probe.py > /tmp/prof$v.log 2>&1
f=$(find /tmp/prof$v -name "*kernel_stats.csv" | head -1)
echo "== att_last=$v $f"; head -12 "$f" | cut -c1-200
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/$tag/s5_attlast${v}_kernel_stats.csv
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/$tag/prof.log
