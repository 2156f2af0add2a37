"""Per-kernel device times (HIP events) of small calls: where a single read's 0.25 ms goes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from catfish_amd.engine import HipEngine
import bench

w = bench.load_weights()
for prec in ("fp32", "bf16"):
    eng = HipEngine(w, device=0, max_windows_per_pass=4096, precision=prec)
    for n in (118, 768, 2048):
        x = torch.randn(n, 35, device="cuda")
        y = torch.empty(n * 35, device="cuda")
        for _ in range(10):
            eng.infer_device(x, out=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.infer_device(x, out=y)
            torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e3
        eng.profile_enable(True, every=1)
        eng.profile_reset()
        for _ in range(20):
            eng.infer_device(x, out=y)
            torch.cuda.synchronize()
        k = eng.profile_read()
        eng.profile_enable(False)
        parts = {name: round(v[0] / 20 * 1e3, 1) for name, v in k.items()}
        print(prec, n, "windows: wall %.3f ms; kernel us per call:" % wall, parts, "sum %.1f us" % sum(parts.values()))
    eng.close()
