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


def profile_only(B, steps, overlap=True):
    """One trainer, ``steps`` steps, and where a step's wall time goes: the host copies of the batch (``load_batch``: numpy -> static
    device buffers), the graph launch, the device time of the replayed graph (HIP events on the stream it runs on) and the
    loss read-back that ends the step.  Prints one JSON line."""
    w = bench.load_weights()
    pool = bench.make_reads(max(8, B // 100 + 1), seed=5).reshape(-1, 35)
    rng = np.random.default_rng(0)
    x = pool[rng.permutation(len(pool))[:B]]
    y = np.repeat((np.arange(B) % 2)[:, None], 35, axis=1).astype(np.float32)
    tr = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=0.8, device="cuda", seed=0)
    tr.step_impl.overlap_wgrad = bool(overlap)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.5:                        # clocks up
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    # the same step taken apart (untimed above): host copies, launch, device, read-back
    sx, b, sloss = tr._static
    parts = {"load_batch_ms": 0.0, "replay_call_ms": 0.0, "graph_device_ms": 0.0, "loss_readback_ms": 0.0}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(steps):
        t = time.perf_counter()
        tr.step_impl.load_batch(b, x, y)
        torch.cuda.synchronize()
        parts["load_batch_ms"] += time.perf_counter() - t
        t = time.perf_counter()
        e0.record()
        tr._graph.replay()
        e1.record()
        parts["replay_call_ms"] += time.perf_counter() - t
        t = time.perf_counter()
        float(sloss.detach())
        parts["loss_readback_ms"] += time.perf_counter() - t
        parts["graph_device_ms"] += e0.elapsed_time(e1) * 1e-3
    print(json.dumps({"batch": B, "steps": steps, "overlap_wgrad": bool(overlap), "ms_per_step": wall * 1e3, "windows_per_s": B / wall,
                      "parts_ms": {k: v / steps * 1e3 for k, v in parts.items()},
                      "what": "parts measured in a second loop with a synchronise after the batch copies: load_batch = numpy -> static device "
                              "buffers (pageable H2D), replay_call = host time of hipGraphLaunch, graph_device = HIP events around the "
                              "replay on its stream, loss_readback = the D2H of the loss that ends a step (waits for the graph)"}))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256, help="windows per training batch (the reference trains with 256)")
    ap.add_argument("--profile-only", action="store_true",
                    help="only the default (native, graph-replayed) trainer: warm-up, then --steps timed steps with a host/device "
                         "breakdown -- the command to put under rocprofv3 --kernel-trace --stats (every traced launch then belongs "
                         "to a whole step: kernel-time sum / steps against wall per step = the launch-gap share)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--no-overlap", action="store_true", help="profile-only: weight gradients on the main stream (round 5's order)")
    args = ap.parse_args()
    B = args.batch
    if args.profile_only:
        return profile_only(B, args.steps, overlap=not args.no_overlap)
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
