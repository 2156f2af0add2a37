"""Diagnostic: per-wave phase timing of the pipelined bf16x3 biGRU mid-layer kernel (gru_bf16x3_pipe_kernel).

Needs a library built with -DCF_X3_STAMP=1 (tools/x3var/libcatfish_x3_stamp_8_8_0.so, see tools/build_x3_variants.sh): every non-LAST
kernel writes, per wave-task, the s_memtime cycles it spent in phases A, B, C of all 35 steps, its total and its start / end
stamps into the dense-partial buffer (the LAST layer writes no partials in such a build, so the mid layer's stamps survive).
usage:  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/x3var/libcatfish_x3_stamp_8_8_0.so python tools/exp_x3_stamps.py [out.npy]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd import _native as N  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402

n_layers = int(os.environ.get("X3_LAYERS", 3))          # 2: the stamps that survive are layer 0's (Cin = 32)
if n_layers == 3:
    w = bench.load_weights()
else:
    from oracle import catfish_oracle as oracle
    w = oracle.random_weights(seed=1, n_layers=n_layers)
if os.environ.get("X3_ZERO_WEIGHTS"):                    # every MFMA operand zero (weights, biases, hence h): the DVFS check --
    w = {k: np.zeros_like(v) for k, v in w.items()}      # same instruction stream, no data toggling; does the clock rise?
    w = {k: (np.ones_like(v) if k.endswith("moving_variance") else v) for k, v in w.items()}
n_win = int(os.environ.get("X3_WINDOWS", 256 * 118))
eng = HipEngine(w, n_layers=n_layers, device=0, max_windows_per_pass=n_win, precision="bf16x3")
x = torch.randn(n_win, 35, device="cuda")
for _ in range(20):
    eng.infer_device(x)
torch.cuda.synchronize()
n_tiles = (n_win + 31) // 32
raw = np.empty(2 * n_tiles * 8 * 2, dtype=np.float32)
N.check(eng._lib.cf_debug_stage(eng._handle, 100, raw.size, raw.ctypes.data_as(C.c_void_p)))
st = raw.view(np.int64).reshape(2, n_tiles, 8)
if len(sys.argv) > 1:
    np.save(sys.argv[1], st)
for d, name in ((0, "fw"), (1, "bw")):
    s = st[d]
    tot = s[:, 4]
    print("%s: wave-tasks %d  total cycles min/median/max %d / %d / %d   phases (median per step): A %.0f B %.0f C %.0f  sum %.0f   prologue %.0f   clock %.2f GHz" % (
        name, len(s), tot.min(), np.median(tot), tot.max(), np.median(s[:, 0]) / 35, np.median(s[:, 1]) / 35, np.median(s[:, 2]) / 35,
        np.median(s[:, :3].sum(1)) / 35, np.median(s[:, 6]), np.median(tot / np.maximum(s[:, 3], 1)) * 0.1))
# wall time of the launches themselves, from the engine's own per-kernel events
eng.profile_enable(True, every=1)
eng.profile_reset()
for _ in range(30):
    eng.infer_device(x)
torch.cuda.synchronize()
print("kernels_ms", {k: round(ms / max(n, 1), 4) for k, (ms, n) in eng.profile_read().items()})
eng.close()
