"""Experiment: fused (one launch, dynamic queues) vs per-layer GRU launches at several batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from catfish_amd.engine import HipEngine
import bench

w = bench.load_weights()
for reads in (64, 128, 256, 512, 1024, 2048):
    nwin = reads * 118
    x = torch.randn(nwin, 35, device="cuda")
    out = torch.empty(nwin * 35, device="cuda")
    for fuse in (False, True):
        eng = HipEngine(w, device=0, max_windows_per_pass=nwin, fuse_layers=fuse)
        for _ in range(3):
            eng.infer_device(x, out=out)
        torch.cuda.synchronize()
        steps = max(4, 8192 // reads)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.infer_device(x, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("reads %5d fuse %d  %.1f M samples/s  %.3f ms/step" % (reads, fuse, steps * reads * 4096 / dt / 1e6, dt / steps * 1e3), flush=True)
        eng.close()
