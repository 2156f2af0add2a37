"""Diagnostic: per-wave phase timing of the pipelined bf16 biGRU mid-layer kernel.

Needs a library built with -DCF_PIPE_STAMP=1 (tools/abl/libcatfish_pipe_stamp.so): every non-LAST kernel writes, per
wave, the s_memtime cycles it spent in phases P1..P4 of all 35 steps, its total and its start/end stamps into the
dense-partial buffer (the LAST layer writes no partials in such a build, so the mid layer's stamps survive).
usage:  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/abl/libcatfish_pipe_stamp.so python tools/exp_pipe_stamps.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402

w = bench.load_weights()
n_win = 256 * 118
eng = HipEngine(w, device=0, max_windows_per_pass=n_win, precision="bf16")
x = torch.randn(n_win, 35, device="cuda")
for _ in range(20):
    eng.infer_device(x)
torch.cuda.synchronize()
n_tiles = (n_win + 31) // 32
raw = np.empty(2 * n_tiles * 8 * 2, dtype=np.float32)
import ctypes as C
from catfish_amd import _native as N
N.check(eng._lib.cf_debug_stage(eng._handle, 100, raw.size, raw.ctypes.data_as(C.c_void_p)))
st = raw.view(np.int64).reshape(2, n_tiles, 8)
if len(sys.argv) > 1:
    np.save(sys.argv[1], st)                     # raw per-wave stamps for offline analysis
for d, name in ((0, "fw"), (1, "bw")):
    s = st[d]
    tot = s[:, 4]
    print("%s: waves %d  total cycles min/median/max %d / %d / %d   phases (median per step): Q1 %.0f Q2 %.0f Q3 %.0f (unused %.0f)  sum %.0f" % (
        name, len(s), tot.min(), np.median(tot), tot.max(), np.median(s[:, 0]) / 35, np.median(s[:, 1]) / 35, np.median(s[:, 2]) / 35,
        np.median(s[:, 3]) / 35, np.median(s[:, :4].sum(1)) / 35))
t_begin = st[:, :, 5].min()
t_end = st[:, :, 6].max()
print("first start -> last end: %d cycles (s_memtime ticks); start spread %d, end spread %d" % (
    t_end - t_begin, st[:, :, 5].max() - t_begin, t_end - st[:, :, 6].min()))
# slowest / fastest waves: which workgroup (blockIdx.x) and wave
order = np.argsort(st[:, :, 4].reshape(-1))
flat = st.reshape(-1, 8)
for k in list(order[:3]) + list(order[-3:]):
    print("  wave tile %5d dir %d  wg %4d wave %d  total %d" % (k % n_tiles, k // n_tiles, flat[k, 7] >> 8, flat[k, 7] & 255, flat[k, 4]))
eng.close()
