"""Per-variable error of three training steps against the reference-graph trajectory (tests/golden/graph_train_golden.npz).

python tools/train_vs_graph.py [cuda-native|cuda-torch|cpu]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from catfish_amd.training import Trainer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cuda-native"
with np.load(os.path.join(ROOT, "tests/golden/graph_train_golden.npz")) as z:
    g = {k: z[k] for k in z.files}
with np.load(os.path.join(ROOT, "tests/golden/ckpnt-30000-inference.npz")) as z:
    w = {k: z[k] for k in z.files}
kw = dict(device="cuda", native=True) if mode == "cuda-native" else \
    dict(device="cuda", native=False, use_graph=False) if mode == "cuda-torch" else dict(device="cpu", native=False, use_graph=False)
tr = Trainer(w, 3, 2, "RMSProp", 1e-3, keep_prob=1.0, **kw)
start = {k: v.detach().clone() for k, v in tr.net.trainable().items()}
for step in range(g["train_x"].shape[0]):
    loss = tr.train_step(g["train_x"][step], g["train_y"][step])
    print("step %d loss %.9f ref %.9f" % (step, loss, g["train32_loss"][step]))
rows = []
for k, p in tr.net.trainable().items():
    d = (p.detach() - start[k]).cpu().numpy()
    ref = g["train32_delta/" + k]
    rows.append((np.abs(d - ref).max() / np.abs(ref).max(), k, np.abs(ref).max(), float(start[k].abs().max())))
for r in sorted(rows, reverse=True)[:12]:
    print("%.3e  %-70s max|delta| %.2e max|p| %.2f" % r)
