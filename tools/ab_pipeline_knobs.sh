#!/bin/bash
# A/B of the ramp-up of a file-driven shard's first batches (sharding.RAMP; CATFISH_RAMP=0 behind the debug switch turns it off) on ONE
# box, interleaved rounds: host-to-host rate for reference, then the CLI leg of bench.py with and without the ramp.
# (profiles/r05_ab_pipeline_knobs.log also holds the stream-priority experiment of round 5, whose knob was removed with the idea.)
export CATFISH_DEBUG_KNOBS=1
for r in 1 2; do
  python tools/bench_e2e.py --batches 96 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r  e2e fp32 %.1f M  %.4f ms/batch' % (d['value']/1e6, d['ms_per_batch']))"
  for ramp in 1 0; do
    CATFISH_RAMP=$ramp python bench.py --no-cpu-baseline --no-extra-precisions --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cli_end_to_end']; print('round $r ramp $ramp  cli %.1f M in %.4f s  infer_s %.4f  sharded_gather %.1f M' % (c['value']/1e6, c['seconds'], c['rank0']['infer_s'], d['sharded_gather']['value']/1e6))"
  done
done
