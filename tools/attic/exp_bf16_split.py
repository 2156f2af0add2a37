"""bf16 forward pass of one 256-read batch as 1, 2, 4 launches of the whole layer stack (smaller launches keep a layer's output
inside the 256 MB Infinity Cache until the next layer has read it; they also fill the chip less)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from catfish_amd.engine import HipEngine
import numpy as np


def main():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with np.load(os.path.join(root, "tests", "golden", "ckpnt-30000-inference.npz")) as z:
        w = {k: z[k] for k in z.files}
    for prec in ("bf16", "fp32"):
        n = 256 * 118
        eng = HipEngine(w, device=0, max_windows_per_pass=n, precision=prec)
        x = torch.randn(n, 35, device="cuda")
        out = torch.empty(n * 35, device="cuda")
        for parts in (1, 2, 4, 8):
            step = (n // parts + 31) // 32 * 32
            cuts = [(a, min(n, a + step)) for a in range(0, n, step)]
            def run():
                for a, b in cuts:
                    eng.infer_device(x[a:b], out=out[a * 35:b * 35])
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            print(json.dumps(dict(precision=prec, launches=parts, windows_per_launch=step, ms=e0.elapsed_time(e1) / 50)), flush=True)
        eng.close()


main()
