#!/bin/bash
# One parametrised script for every gpurun call of a round (replaces the per-experiment scripts of round 3):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# steps (each bounded by its own `timeout`, each writes under gpurun_out/<tag>/):
#   tests[:<pytest -k expr>]  pytest -m gpu (optionally filtered)
#   bench                     the driver's headline command
#   stats                     rocprofv3 --kernel-trace --stats of the headline command -> bench_kernel_stats.csv
#   cfg4stats | secstats      the same for `bench.py --config cfg4` / `bench.py --secondary-only`
#   pmc | pmc4                FETCH_SIZE / WRITE_SIZE / SQ passes over tools/pmc_probe.py (cfg2 / cfg4) -> pmc_traffic*.json
#   pmcdecode                 the three counter passes over the configs[2] decode (tools/pm_timing.py) -> pmc_decode_summary.txt
#   pmcsr                     the same over the configs[4] SampleRNN generation (tools/sr_timing.py) -> pmc_sr_summary.txt
#   run:<command>             anything else, logged to run_<n>.log
tag=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
n=0
stats() {  # stats <name> <timeout> <cmd...>
  name=$1; to=$2; shift 2
  timeout $to rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -- "$@" > $out/${name}_under_rocprof.json 2> $out/${name}_rocprof.err
  find /tmp/rp_$name -name "*kernel_stats.csv" -exec cp {} $out/${name}_kernel_stats.csv \;
  rm -rf /tmp/rp_$name
  head -12 $out/${name}_kernel_stats.csv | cut -c1-150
}
pmc() {  # pmc <suffix> <probe cfg>
  suf=$1; cfg=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    PROBE_CFG=$cfg timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python tools/pmc_probe.py > $out/pmc_${c}$suf.log 2>&1
    find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $out/$c$suf.csv \;
    rm -rf /tmp/pmc_$c
    tail -2 $out/pmc_${c}$suf.log
  done
  python tools/pmc_traffic.py $out/FETCH_SIZE$suf.csv $out/WRITE_SIZE$suf.csv $out/pmc_traffic$suf.json "session $tag: tools/pmc_probe.py, one training step of the $cfg shapes at T_dec = 100, eager launches, under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"
  PROBE_CFG=$cfg timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc_sq -- python tools/pmc_probe.py > $out/pmc_SQ$suf.log 2>&1
  find /tmp/pmc_sq -name "*counter_collection.csv" -exec cp {} $out/SQ$suf.csv \;
  rm -rf /tmp/pmc_sq
  python tools/pmc_summary.py $out/SQ$suf.csv > $out/pmc_sq_summary$suf.txt 2>&1
  rm -f $out/FETCH_SIZE$suf.csv $out/WRITE_SIZE$suf.csv $out/SQ$suf.csv
  head -40 $out/pmc_sq_summary$suf.txt
}
pmcx() {  # pmcx <name> <cmd...>: FETCH_SIZE / WRITE_SIZE / SQ passes over any command, per-kernel averages of each
  name=$1; shift
  : > $out/pmc_${name}_summary.txt
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcx -- "$@" > $out/pmc_${name}.log 2>&1
    find /tmp/pmcx -name "*counter_collection.csv" -exec cp {} /tmp/pmcx.csv \;
    python tools/pmc_summary.py /tmp/pmcx.csv >> $out/pmc_${name}_summary.txt 2>&1
    rm -rf /tmp/pmcx /tmp/pmcx.csv
  done
  grep -A12 "pm_kernel\|srp_kernel" $out/pmc_${name}_summary.txt | head -60
}
for step in "$@"; do
  case "$step" in
    tests)   timeout 3000 python -m pytest tests -q -m gpu -x --timeout 600 2>&1 | tail -25 | tee $out/tests.log ;;
    tests:*) timeout 2700 python -m pytest tests -q -m gpu --timeout 600 -k "${step#tests:}" 2>&1 | tail -40 | tee -a $out/tests_k.log ;;
    bench)   ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time
             tail -c 3000 $out/bench.json; tail -3 $out/bench.time ;;
    stats)   stats bench 600 python bench.py --no-cpu-baseline --no-parity --no-secondary --no-f32-gemm ;;
    cfg4stats) stats cfg4 600 python bench.py --config cfg4 --no-cpu-baseline --no-dense --no-parity --no-secondary --steps 3 --warmup 1 ;;
    secstats) stats secondary 600 python bench.py --secondary-only ;;
    pmc)     pmc "" cfg2 ;;
    pmc4)    pmc _cfg4 cfg4 ;;
    pmcdecode) MODE=decode pmcx decode python tools/pm_timing.py ;;
    pmcsr)   pmcx sr python tools/sr_timing.py ;;
    run:*)   n=$((n+1)); ( eval "${step#run:}" ) > $out/run_$n.log 2>&1; tail -30 $out/run_$n.log ;;
    *) echo "unknown step $step" ;;
  esac
done
