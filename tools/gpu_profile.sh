#!/bin/bash
# rocprofv3 kernel stats of the headline command (and of the secondary benches) -> gpurun_out/<tag>/
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh r02z'
tag=${1:-prof}
mkdir -p gpurun_out/$tag
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p1 -- python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/p1.err
find gpurun_out/$tag/p1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/bench_kernel_stats.csv \;
rm -rf gpurun_out/$tag/p1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/p2 -- python bench.py --config cfg4 --no-cpu-baseline --no-dense --steps 2 --warmup 1 > gpurun_out/$tag/cfg4_under_rocprof.json 2> gpurun_out/$tag/p2.err
find gpurun_out/$tag/p2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/cfg4_kernel_stats.csv \;
rm -rf gpurun_out/$tag/p2
head -8 gpurun_out/$tag/bench_kernel_stats.csv | cut -c1-160; head -6 gpurun_out/$tag/cfg4_kernel_stats.csv | cut -c1-160
