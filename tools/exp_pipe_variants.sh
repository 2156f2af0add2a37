#!/bin/bash
# A/B of the pipelined bf16 biGRU kernel's vector work (CF_PIPE_VAR bits, gru_bf16_pipe.hpp): builds one library per variant
# on the GPU box (into gpurun_out/abl/, removed afterwards) and times the bf16 step with each; results stay bit-identical.
# usage: bash tools/exp_pipe_variants.sh 0 1 2 3 4 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abl
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -DCF_PIPE_VAR=$v -o gpurun_out/abl/libvar$v.so catfish_amd/csrc/catfish_hip.hip 2>gpurun_out/abl/build$v.log &
done
wait
for rep in 1 2; do
for v in "$@"; do
  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=$PWD/gpurun_out/abl/libvar$v.so python bench.py --precision bf16 --no-cpu-baseline --no-extra-precisions --no-sharded-leg --steps 40 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels_ms']
print('var $v  step %.4f ms  first %.1f mid %.1f last %.1f us  max_dp %.3g match %.5f' % (r['ms_per_step'], k['gru_layer_first']*1e3, k['gru_layer_mid']*1e3, k['gru_layer_last']*1e3, r['parity']['max_abs_dp_vs_fp64_oracle'], r['parity']['label_match_vs_fp32_oracle']))"
done
done
rm -rf gpurun_out/abl
