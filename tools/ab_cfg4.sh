#!/bin/bash
# A/B of the round-5 configs[3] changes on ONE box (box-to-box spread is ~1 %): bf16-in readout products and bf16 gradient
# copies written by the backward tick, each on / off, two passes.
for pass in 1 2; do
  for v in "1 1" "0 1" "1 0" "0 0"; do
    set -- $v
    ms=$(PARROT_BF16_READOUT=$1 PARROT_BF16_DG16=$2 python bench.py --config cfg4 --no-cpu-baseline --no-dense --no-parity --no-secondary --no-roofline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "pass $pass readout_bf16in=$1 dG16_from_scan=$2 ms_per_step=$ms"
  done
done
