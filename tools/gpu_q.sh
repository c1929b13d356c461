#!/bin/bash
mkdir -p gpurun_out/r02q
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r02q/$name -- python bench.py --config cfg4 --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-dense > gpurun_out/r02q/$name.json 2> gpurun_out/r02q/$name.err
  f=$(find gpurun_out/r02q/$name -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $f wk_kernel > gpurun_out/r02q/$name.txt 2>&1; rm -rf gpurun_out/r02q/$name; cat gpurun_out/r02q/$name.txt; }
run p1 FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run p2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
