#!/bin/bash
# Build timing-only ablation variants of the library into build/ (results are wrong; timing only).
set -e
cd "$(dirname "$0")/.."
mkdir -p build
for A in ${ABLATIONS:-0 1 2 3 4}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCF_ABLATE=$A -o build/libcatfish_hip_abl$A.so catfish_amd/csrc/catfish_hip.hip
done
ls -la build/*.so
