#!/bin/bash
# Round-2 evidence run: full GPU suite, bench (default + persistent forward), rocprofv3 kernel stats, PMC traffic
# passes (separate runs, no tracing combined with --pmc), secondary measurements.
mkdir -p gpurun_out/r02h
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r02h/all_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r02h/all_tests.log
tail -n 6 gpurun_out/r02h/all_tests.log
timeout 600 python bench.py > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err; tail -c 400 gpurun_out/r02h/bench.json
PARROT_SCHEDULE=4 timeout 300 python bench.py --no-cpu-baseline --no-dense > gpurun_out/r02h/bench_sched4.json 2> gpurun_out/r02h/bench_sched4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02h/prof -o b -- python bench.py --no-cpu-baseline --no-dense > gpurun_out/r02h/bench_under_rocprof.json 2> gpurun_out/r02h/prof.err
find gpurun_out/r02h/prof -name "*kernel_trace.csv" -delete
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02h/pmc_f -o f -- python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-dense > /dev/null 2> gpurun_out/r02h/pmc_f.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02h/pmc_w -o w -- python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-dense > /dev/null 2> gpurun_out/r02h/pmc_w.err
F=$(find gpurun_out/r02h/pmc_f -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r02h/pmc_w -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/r02h/pmc_traffic.json
rm -rf gpurun_out/r02h/pmc_f gpurun_out/r02h/pmc_w
timeout 600 python tools/bench_extra.py 2>/dev/null | tail -1 > gpurun_out/r02h/secondary.json; cat gpurun_out/r02h/secondary.json
PARROT_SAMPLE_PERSIST=0 timeout 600 python tools/bench_extra.py 2>/dev/null | tail -1 > gpurun_out/r02h/secondary_launches.json
