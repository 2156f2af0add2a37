"""BASELINE config 5: train step (forward + backward + Adam), windows/s.

Synthetic config-2 windows, per-window uniform labels, 128 positive + 128 negative per batch of
256 (mimics ExampleDb.get_training_set, networks/trainingDB/ExampleDb.py:50-83), keep_prob 0.8,
lr 1e-3.  Also prints the 10-step loss trajectory next to a CPU run with dropout off.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from catfish_amd.training import Trainer  # noqa: E402
import bench  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256, help="windows per training batch (the reference trains with 256)")
    args = ap.parse_args()
    B = args.batch
    w = bench.load_weights()
    reads = bench.make_reads(max(8, B // 100 + 1), seed=5).reshape(-1, 35)
    rng = np.random.default_rng(0)
    x = reads[rng.permutation(len(reads))[:B]]
    y = np.repeat((np.arange(B) % 2)[:, None], 35, axis=1).astype(np.float32)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    gpu = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=1.0, device=dev, seed=0)     # native HIP biGRU kernels on a GPU
    cpu = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=1.0, device="cpu", seed=0)
    lg = [gpu.train_step(x, y) for _ in range(10)]
    lc = [cpu.train_step(x, y) for _ in range(10)]
    tr = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=0.8, device=dev, seed=0)
    for _ in range(3):
        tr.train_step(x, y)
    if dev == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        tr.train_step(x, y)
    if dev == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    modes = {}
    if dev == "cuda":
        for name, kw in (("torch_eager", dict(native=False, use_graph=False)), ("torch_hipgraph", dict(native=False, use_graph=True)),
                         ("native_eager", dict(native=True, use_graph=False))):
            t2 = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=0.8, device=dev, seed=0, **kw)
            for _ in range(3):
                t2.train_step(x, y)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                t2.train_step(x, y)
            torch.cuda.synchronize()
            modes[name + "_ms_per_step"] = (time.perf_counter() - t1) / 5 * 1e3
    print(json.dumps({"metric": "training windows/s (config 5: the whole step on HIP kernels through the C ABI -- conv stack, biGRU fwd/bwd/wgrad with in-kernel dropout, dense head + loss, fused TF-style Adam + re-tiling; no autograd)",
                      "device": dev, "native": bool(getattr(tr, "native", False)), "other_modes": modes,
                      "batch": B, "value": n * B / dt, "ms_per_step": dt / n * 1e3,
                      "loss_10_steps_device": lg, "loss_10_steps_cpu": lc,
                      "max_loss_diff": float(np.max(np.abs(np.array(lg) - np.array(lc))))}))


if __name__ == "__main__":
    main()
