"""Build the HIP C-ABI library in-tree: catfish_amd/csrc/libcatfish_hip.so.

``hipcc --offload-arch=gfx950`` cross-compiles without a GPU.  The built .so is
git-ignored but travels with the working tree to the GPU box.
"""
from __future__ import annotations

import json
import os
import re
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


RESOURCES = os.path.join(_HERE, "csrc", "kernel_resources.json")
_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|"
                     r"VGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)")
_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
         "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
         "LDS Size [bytes/block]": "lds_bytes"}


def parse_resource_remarks(text):
    """``-Rpass-analysis=kernel-resource-usage`` output -> {mangled kernel name: {vgprs, agprs, scratch_bytes_per_lane, ...}}."""
    out, cur = {}, None
    for m in _REMARK.finditer(text):
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None:
            try:
                cur[_KEYS[key]] = int(val)
            except ValueError:
                pass
    return out


def demangle(names):
    """Mangled -> readable kernel names through llvm-cxxfilt / c++filt when one is there (the record stays usable without)."""
    tool = next((t for t in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("c++filt")) if t and os.path.exists(t)), None)
    names = list(names)
    if tool is None or not names:
        return dict(zip(names, names))
    res = subprocess.run([tool], input="\n".join(names) + "\n", stdout=subprocess.PIPE, universal_newlines=True)
    got = res.stdout.splitlines()
    return dict(zip(names, got)) if res.returncode == 0 and len(got) == len(names) else dict(zip(names, names))


def check_resources(resources):
    """What the design of the kernels relies on and a compiler change could silently break: NO kernel may touch scratch memory or
    spill vector registers (a spilled 470-value biGRU kernel would still be correct, at a fraction of its speed), and with
    -amdgpu-mfma-vgpr-form no kernel may hold MFMA accumulators in AccVGPRs.  -> list of complaints (empty = fine)."""
    bad = []
    for name, r in sorted(resources.items()):
        if r.get("scratch_bytes_per_lane", 0) != 0 or r.get("vgpr_spills", 0) != 0:
            bad.append("%s: scratch %s B/lane, %s VGPR spills" % (name, r.get("scratch_bytes_per_lane"), r.get("vgpr_spills")))
        if r.get("vgprs", 0) + r.get("agprs", 0) > 512:
            bad.append("%s: %s VGPRs + %s AGPRs exceed the 512-register file" % (name, r.get("vgprs"), r.get("agprs")))
    return bad


def build_native(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs instead of AccVGPRs.  The conv kernels feed every MFMA result through VALU
    # (ReLU) and back in as a B operand: the AccVGPR form costs them 72 v_accvgpr_read per position (res_stack2_kernel
    # 0.205 -> 0.198 ms; the fp32 biGRU kernels are unchanged).  gru_bf16x3_pipe_kernel is designed around the WHOLE unified
    # 512-register file of a one-wave-per-SIMD launch (launch_bounds(256, 1), ~470 live values, MFMA C/D in arch VGPRs): the
    # remarks below are parsed into csrc/kernel_resources.json and the build FAILS when any kernel uses scratch or spills
    # vector registers (check_resources); tests/test_host_logic.py reads the same record, DESIGN.md section 4 quotes it.
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-Rpass-analysis=kernel-resource-usage", "-o", OUT + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if res.returncode != 0:
        if os.path.exists(OUT + ".tmp"):
            os.unlink(OUT + ".tmp")
        raise RuntimeError("hipcc failed:\n" + "\n".join(l for l in res.stdout.splitlines() if "kernel-resource-usage" not in l))
    resources = parse_resource_remarks(res.stdout)
    names = demangle(resources)
    record = {"flags": cmd[1:-3], "kernels": {names[k]: v for k, v in sorted(resources.items())}}
    bad = check_resources(record["kernels"])
    if bad or not resources:
        os.unlink(OUT + ".tmp")
        raise RuntimeError("hipcc built kernels that spill (or reported no kernel resources at all):\n  " + "\n  ".join(bad))
    with open(RESOURCES, "w") as fh:
        json.dump(record, fh, indent=1, sort_keys=True)
        fh.write("\n")
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
