"""Experiment: alternate full batches over E engines on E torch streams (tail overlap across steps)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from catfish_amd.engine import HipEngine
import bench

def run(n_eng, steps=40, reads=256, inner_streams=1):
    w = bench.load_weights()
    nwin = reads * 118
    engs = [HipEngine(w, device=0, max_windows_per_pass=nwin, n_streams=inner_streams) for _ in range(n_eng)]
    streams = [torch.cuda.Stream() for _ in range(n_eng)]
    x = torch.randn(nwin, 35, device="cuda")
    outs = [torch.empty(nwin * 35, device="cuda") for _ in range(n_eng)]
    for i in range(3 * n_eng):
        with torch.cuda.stream(streams[i % n_eng]):
            engs[i % n_eng].infer_device(x, out=outs[i % n_eng])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % n_eng]):
            engs[i % n_eng].infer_device(x, out=outs[i % n_eng])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in engs: e.close()
    return steps * reads * 4096 / dt / 1e6, dt / steps * 1e3

if __name__ == "__main__":
    for reads in (256, 1024):
        for n_eng in (1, 2, 3, 1, 2):
            v, ms = run(n_eng, steps=40 if reads == 256 else 12, reads=reads)
            print("reads", reads, "engines", n_eng, "Msamples/s %.1f" % v, "ms/step %.3f" % ms, flush=True)
