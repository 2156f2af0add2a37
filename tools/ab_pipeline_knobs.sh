#!/bin/bash
# A/B of the host pipeline's two scheduling choices on ONE box, interleaved: the compute stream's priority (CATFISH_PIPE_PRIO) and the
# ramp-up of a file-driven shard's first batches (CATFISH_RAMP); prints the host-to-host rate and the CLI leg for every combination.
export CATFISH_DEBUG_KNOBS=1
for r in 1 2; do   # (CATFISH_PIPE_PRIO was a knob of the round-5 experiment only; the product has no stream priorities)
  for prio in 1 0; do
    CATFISH_PIPE_PRIO=$prio python tools/bench_e2e.py --batches 96 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r prio $prio  e2e fp32 %.1f M  %.4f ms/batch' % (d['value']/1e6, d['ms_per_batch']))"
    CATFISH_PIPE_PRIO=$prio python tools/bench_e2e.py --precision bf16 --batches 96 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r prio $prio  e2e bf16 %.1f M  %.4f ms/batch' % (d['value']/1e6, d['ms_per_batch']))"
    for ramp in 1 0; do
      CATFISH_PIPE_PRIO=$prio CATFISH_RAMP=$ramp python bench.py --no-cpu-baseline --no-extra-precisions --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cli_end_to_end']; print('round $r prio $prio ramp $ramp  cli %.1f M in %.4f s  infer_s %.4f  sharded_gather %.1f M' % (c['value']/1e6, c['seconds'], c['rank0']['infer_s'], d['sharded_gather']['value']/1e6))"
    done
  done
done
