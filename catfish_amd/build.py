"""Build the HIP C-ABI library in-tree: catfish_amd/csrc/libcatfish_hip.so.

``hipcc --offload-arch=gfx950`` cross-compiles without a GPU.  The built .so is
git-ignored but travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "catfish_hip.hip")
OUT = os.path.join(_HERE, "csrc", "libcatfish_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "catfish_hip.h")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    csrc = os.path.dirname(SRC)
    deps = [HEADER] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".h"))]
    newest = max(os.path.getmtime(p) for p in deps)
    return os.path.getmtime(OUT) < newest


def build_native(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs instead of AccVGPRs.  No kernel here needs more than 256 registers, and
    # the conv kernels feed every MFMA result through VALU (ReLU) and back in as a B operand: the AccVGPR form costs them 72
    # v_accvgpr_read per position (res_stack2_kernel 0.205 -> 0.198 ms; the biGRU kernels are unchanged).
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    return OUT


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
