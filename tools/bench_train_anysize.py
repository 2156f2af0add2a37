"""Training step at geometries other than the shipped 64 / 32 (draws of networks/train_validate.py:66-111): torch autograd around
the any-size HIP recurrence kernels (catfish_amd/anysize_train.py) against the pure torch-autograd restatement, both on the GPU,
eager and under HIP-graph replay; loss trajectories compared."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from catfish_amd.training import Trainer  # noqa: E402
from oracle import catfish_oracle as oracle  # noqa: E402
import bench  # noqa: E402


def timed(tr, x, y, n):
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = 256
    reads = bench.make_reads(8, seed=5).reshape(-1, 35)
    rng = np.random.default_rng(0)
    x = reads[rng.permutation(len(reads))[:B]]
    y = np.repeat((np.arange(B) % 2)[:, None], 35, axis=1).astype(np.float32)
    for h, c, nl, nr in ((32, 16, 3, 2), (128, 64, 3, 2), (256, 128, 3, 2), (128, 0, 2, 0)):
        w = oracle.random_weights(seed=3, layer_size=h, n_layers=nl, layer_size_res=max(c, 16), n_layers_res=nr)
        a = Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=1.0, seed=0)
        b = Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=1.0, seed=0, native=False)
        la = [a.train_step(x, y) for _ in range(8)]
        lb = [b.train_step(x, y) for _ in range(8)]
        res = dict(layer_size=h, layer_size_res=c, n_layers=nl, n_layers_res=nr, batch=B, anysize=bool(a.anysize),
                   max_loss_diff_vs_torch=float(np.max(np.abs(np.array(la) - np.array(lb)))))
        res["anysize_hipgraph_ms"] = timed(Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=0.8, seed=0), x, y, 10)
        res["anysize_eager_ms"] = timed(Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=0.8, seed=0, use_graph=False), x, y, 5)
        res["torch_hipgraph_ms"] = timed(Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=0.8, seed=0, native=False), x, y, 3)
        res["torch_eager_ms"] = timed(Trainer(w, nl, nr, "Adam", 1e-3, keep_prob=0.8, seed=0, native=False, use_graph=False), x, y, 3)
        print(json.dumps(res), flush=True)


main()
