"""Diagnostic: segment timing of the latency-mode fp32 biGRU kernel (library built with -DCF_COOP_STAMP=1).
usage:  CATFISH_DEBUG_KNOBS=1 CATFISH_HIP_LIB=tools/abl/libcatfish_coop_stamp.so python tools/exp_coop_stamps.py [n_windows]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd import _native as N  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 118
w = bench.load_weights()
eng = HipEngine(w, device=0, max_windows_per_pass=4096, n_layers=2)     # layer 0 non-LAST (Cin 32) only... use 3 layers below
eng.close()
eng = HipEngine(w, device=0, max_windows_per_pass=4096)
x = torch.randn(n_win, 35, device="cuda")
for _ in range(20):
    eng.infer_device(x)
torch.cuda.synchronize()
n_tiles = (n_win + 15) // 16
raw = np.empty(2 * n_tiles * 8 * 2, dtype=np.float32)
N.check(eng._lib.cf_debug_stage(eng._handle, 100, raw.size, raw.ctypes.data_as(C.c_void_p)))
st = raw.view(np.int64).reshape(2, n_tiles, 8)
s = st.reshape(-1, 8)
print("%d windows: median cycles per step: h read + r/u MFMAs %.0f | r, r*h, barrier %.0f | c MFMAs %.0f | u, c, h', barrier %.0f | total/35 %.0f" % (
    n_win, np.median(s[:, 0]) / 35, np.median(s[:, 1]) / 35, np.median(s[:, 2]) / 35, np.median(s[:, 3]) / 35, np.median(s[:, 4]) / 35))
eng.close()
