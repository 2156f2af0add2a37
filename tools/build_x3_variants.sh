#!/bin/bash
# Diagnostic builds of the pipelined bf16x3 biGRU kernel -> tools/x3var/ (travels to the GPU box, unlike tools/abl/):
#   stamp[:XUA:XCA[:NT]]  -DCF_X3_STAMP=1 per-phase s_memtime stamps (tools/exp_x3_stamps.py), optionally another split of the x
#                         products over the phases (CF_X3_XUA / CF_X3_XCA) or non-temporal hints (CF_X3_NT)  -> libcatfish_x3_stamp_XUA_XCA_NT.so
#   abl:N                 -DCF_X3_ABL=N timing only, WRONG results: 1 no vector work, 2 no MFMAs, 4 no A-ring refills -> libcatfish_x3_ablN.so
# usage: tools/build_x3_variants.sh stamp stamp:12:4 abl:1 ...
# Load one with CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/x3var/<lib> python tools/exp_x3_stamps.py | bench.py --precision bf16x3 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/x3var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form"
for v in "$@"; do
  IFS=: read -r kind a b c <<< "$v"
  if [ "$kind" = stamp ]; then
    hipcc $F -DCF_X3_STAMP=1 -DCF_X3_XUA=${a:-8} -DCF_X3_XCA=${b:-8} -DCF_X3_NT=${c:-0} -o tools/x3var/libcatfish_x3_stamp_${a:-8}_${b:-8}_${c:-0}.so catfish_amd/csrc/catfish_hip.hip &
  else
    hipcc $F -DCF_X3_ABL=$a -o tools/x3var/libcatfish_x3_abl$a.so catfish_amd/csrc/catfish_hip.hip &
  fi
done
wait
ls -la tools/x3var
