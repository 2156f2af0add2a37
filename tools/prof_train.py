import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from catfish_amd.training import Trainer
import bench
w = bench.load_weights()
reads = bench.make_reads(8, seed=5).reshape(-1, 35)
x = reads[:256]
y = np.repeat((np.arange(256) % 2)[:, None], 35, axis=1).astype(np.float32)
tr = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=0.8, device="cuda", seed=0, use_graph=False)  # eager so that rocprof sees the kernels
for _ in range(60):
    tr.train_step(x, y)
torch.cuda.synchronize()
