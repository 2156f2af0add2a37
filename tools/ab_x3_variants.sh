#!/bin/bash
# A/B of bf16x3 library variants on ONE box, interleaved rounds (the boxes differ by several percent, and so do clocks over time):
#   usage: bash tools/ab_x3_variants.sh <rounds> <lib or "default"> ...
# prints ms per step and the GRU layer times of `bench.py --precision bf16x3` for every (round, variant).
R=$1; shift
export CATFISH_DEBUG_KNOBS=1
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CATFISH_HIP_LIB; else export CATFISH_HIP_LIB=$v; fi
    python bench.py --precision bf16x3 --no-extra-precisions --no-sharded-leg --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; print('round $r %-40s ms/step %.4f  first %.4f mid %.4f last %.4f res %.4f  max|dp| %.2e' % ('$v', d['ms_per_step'], k['gru_layer_first'], k['gru_layer_mid'], k['gru_layer_last'], k['res_stack2'], d['parity']['max_abs_dp_vs_fp64_oracle']))"
  done
done
