#!/bin/bash
# GPU box: time the bf16 bench with every ablation variant in tools/abl/ (kernel times from the live HIP events)
cd "$GRAFT_REPO_ROOT"
for f in catfish_amd/csrc/libcatfish_hip.so tools/abl/*.so; do
  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=$PWD/$f python bench.py --precision bf16 --no-cpu-baseline --no-extra-precisions --no-sharded-leg --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels_ms']
print('%-40s step %.3f ms  gru0 %.1f mid %.1f last %.1f us' % ('$f'.split('/')[-1], d['ms_per_step'], k['gru_layer_first']*1e3, k['gru_layer_mid']*1e3, k['gru_layer_last']*1e3))
"
done
