"""DVFS check per precision (MI355X_MICROARCH.md, 'DVFS give-back'): the same launches with real weights on random input, and
with every weight and bias zero -- all activations and every MFMA operand zero then, the instruction streams unchanged.  A kernel
that gets faster on zeros is held back by the clock the chip grants under its data-dependent power, not by its schedule.
usage: python tools/exp_zero_operands.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402

real = bench.load_weights()
zero = {k: (np.ones_like(v) if k.endswith("moving_variance") else np.zeros_like(v)) for k, v in real.items()}
n_win = 256 * 118
x = torch.randn(n_win, 35, device="cuda")
for prec in ("fp32", "bf16x3", "bf16"):
    rows = {}
    for name, w in (("random data", real), ("zero operands", zero)):
        eng = HipEngine(w, device=0, max_windows_per_pass=n_win, precision=prec)
        for _ in range(30):
            eng.infer_device(x)
        torch.cuda.synchronize()
        eng.profile_enable(True, every=1)
        eng.profile_reset()
        for _ in range(40):
            eng.infer_device(x)
        torch.cuda.synchronize()
        rows[name] = {k: ms / max(n, 1) for k, (ms, n) in eng.profile_read().items()}
        eng.close()
    for k in ("res_stack2", "gru_layer_first", "gru_layer_mid", "gru_layer_last"):
        a, b = rows["random data"].get(k), rows["zero operands"].get(k)
        if a and b:
            print("%-7s %-16s random data %.4f ms   zero operands %.4f ms   ratio %.2f" % (prec, k, a, b, a / b))
