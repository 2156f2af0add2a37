#!/bin/bash
# Timing-only ablation variants of the pipelined bf16 biGRU kernel -> tools/abl/libcatfish_pipe_abl<bits>.so
# (CF_PIPE_ABL bits: 1 no activation arithmetic, 2 no MFMA, 4 no A-fragment LDS reads, 8 no global loads/stores).
# Load one with CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/abl/libcatfish_pipe_ablN.so python bench.py --precision bf16 ... (results are wrong by construction)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for b in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCF_PIPE_ABL=$b -o tools/abl/libcatfish_pipe_abl$b.so catfish_amd/csrc/catfish_hip.hip &
done
wait
ls -la tools/abl
