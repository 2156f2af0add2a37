"""bf16x3: the pipelined kernel against the round-1 kernel (CATFISH_BF16_PIPE=0) -- error statistics against the fp64 oracle
on the same windows, and the size of the difference between the two.
usage:  CATFISH_DEBUG_KNOBS=1 python tools/exp_x3_accuracy.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402
from oracle import catfish_oracle as oracle  # noqa: E402

os.environ["CATFISH_DEBUG_KNOBS"] = "1"
for name, w in (("ckpnt-30000", bench.load_weights()), ("random seed 3", oracle.random_weights(seed=3))):
    x = np.random.default_rng(1).normal(0, 1.4, size=(1500, 35)).astype(np.float32)
    want = oracle.forward(x, w, np.float64)
    eng = HipEngine(w, device=0, max_windows_per_pass=4096, precision="bf16x3")
    out = {}
    for knob in ("1", "0"):
        os.environ["CATFISH_BF16_PIPE"] = knob
        out[knob] = eng.infer_host(x).astype(np.float64)
    eng.close()
    for knob, label in (("1", "pipelined"), ("0", "round-1  ")):
        e = out[knob] - want
        print("%-14s %s  max|dp| %.3e  rms %.3e  mean %.3e" % (name, label, np.abs(e).max(), np.sqrt((e * e).mean()), e.mean()))
    d = out["1"] - out["0"]
    print("%-14s pipelined - round-1: max %.3e rms %.3e, identical samples %.4f" % (name, np.abs(d).max(), np.sqrt((d * d).mean()), (d == 0).mean()))
