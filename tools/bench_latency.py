"""Latency of small calls: one read (118 windows) and a 256-window micro-batch, device-resident."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from catfish_amd.engine import HipEngine
import bench

w = bench.load_weights()
out = {}
for prec in ("fp32", "bf16x3", "bf16"):
    eng = HipEngine(w, device=0, max_windows_per_pass=4096, precision=prec)
    for n in (118, 256, 512, 768, 1024, 2048, 4096):
        x = torch.randn(n, 35, device="cuda")
        y = torch.empty(n * 35, device="cuda")
        for _ in range(5):
            eng.infer_device(x, out=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            eng.infer_device(x, out=y)
            torch.cuda.synchronize()
        out["%s_%d_windows_ms" % (prec, n)] = (time.perf_counter() - t0) / reps * 1e3
    if prec == "fp32":
        import numpy as np
        xh = np.random.default_rng(0).normal(size=(118, 35, 1))
        for _ in range(5):
            eng.infer_host(xh)
        t0 = time.perf_counter()
        for _ in range(50):
            eng.infer_host(xh)
        out["fp32_118_windows_host_numpy_in_out_ms"] = (time.perf_counter() - t0) / 50 * 1e3
    eng.close()
print(json.dumps(out))
