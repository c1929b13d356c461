#!/bin/bash
# headline bench (as the driver runs it) + rocprofv3 kernel stats of the same step + PMC traffic passes
tag=${1:-r03bench}
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err ) 2> gpurun_out/$tag/bench.time
tail -c 1500 gpurun_out/$tag/bench.json; tail -3 gpurun_out/$tag/bench.time
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p1 -- python bench.py --no-cpu-baseline --no-parity --no-secondary > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/p1.err
find gpurun_out/$tag/p1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/bench_kernel_stats.csv \;
rm -rf gpurun_out/$tag/p1
head -14 gpurun_out/$tag/bench_kernel_stats.csv | cut -c1-170
if [ "$2" = "pmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  # short probe: one training step at T_dec = 100 (~1200 launches, ~35 s per pass); the 800-frame probe does not finish
  PROBE_T=100 PROBE_STEPS=1 PROBE_SCANS=0 timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag/pmc_$c -- python tools/host_launch_probe.py > gpurun_out/$tag/pmc_$c.log 2>&1
  find gpurun_out/$tag/pmc_$c -name "*counter_collection.csv" -exec cp {} gpurun_out/$tag/$c.csv \;
  rm -rf gpurun_out/$tag/pmc_$c
done
python tools/pmc_traffic.py gpurun_out/$tag/FETCH_SIZE.csv gpurun_out/$tag/WRITE_SIZE.csv gpurun_out/$tag/pmc_traffic.json "session $tag: tools/host_launch_probe.py, one training step of the cfg2 shapes at T_dec = 100, under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"
rm -f gpurun_out/$tag/FETCH_SIZE.csv gpurun_out/$tag/WRITE_SIZE.csv
fi
