"""Where the native training step's time goes: graph-replayed timings of its parts (256 windows by default).

python tools/exp_train_parts.py [batch]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from catfish_amd.training import Trainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = bench.load_weights()
reads = bench.make_reads(max(8, B // 100 + 1), seed=5).reshape(-1, 35)
x = torch.from_numpy(reads[:B]).cuda()
y = torch.from_numpy(np.repeat((np.arange(B) % 2)[:, None], 35, axis=1).astype(np.float32)).cuda()
tr = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=0.8, device="cuda", seed=0, use_graph=False)
net, opt, eng = tr.net, tr.opt, tr.engine
opt.keep_grads = True


def timed(name, fn, iters=200):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    print("%-46s %.3f ms" % (name, (time.perf_counter() - t0) / iters * 1e3))


def zero():
    for p in net.trainable().values():
        p.grad = None


def full():
    zero()
    net.loss(x, y, 0.8, None, eng).backward()
    opt.step()


def fwd_bwd():
    zero()
    net.loss(x, y, 0.8, None, eng).backward()


def fwd_only():
    with torch.no_grad():
        net.loss(x, y, 0.8, None, eng)


from catfish_amd.native_train import native_gru_stack  # noqa: E402
pre = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell"
plist = [net.params[(pre % (layer, d)) + k] for layer in range(3) for d in ("fw", "bw")
         for k in ("/gates/kernel", "/gates/bias", "/candidate/kernel", "/candidate/bias")]
xin = torch.randn(B, 35, 32, device="cuda", requires_grad=True)


def gru_only():
    zero()
    xin.grad = None
    native_gru_stack(xin, plist, eng, 0.8).sum().backward()


def res_only():
    zero()
    p = net.params
    a = x[:, None, :]
    for d in range(2):
        j0 = 4 * d
        sc = net._conv_bn(a, j0)
        o = torch.relu(net._conv_bn(a, j0 + 1))
        o = torch.relu(net._conv_bn(o, j0 + 2))
        o = torch.relu(net._conv_bn(o, j0 + 3))
        a = torch.relu(o + sc)
    a.sum().backward()


hin = torch.randn(B, 35, 128, device="cuda", requires_grad=True)


def head_only():
    zero()
    hin.grad = None
    p = net.params
    z = (hin.reshape(-1, 128) @ p["final_fully_connected/kernel"] + p["final_fully_connected/bias"]).reshape(B, 35)
    torch.nn.functional.binary_cross_entropy_with_logits(z, y, reduction="mean").backward()


def opt_only():
    opt.step()


timed("full step (fwd + bwd + Adam)", full)
timed("fwd + bwd", fwd_bwd)
timed("fwd only", fwd_only)
timed("biGRU stack fwd + bwd (native)", gru_only)
timed("residual blocks fwd + bwd (torch)", res_only)
timed("dense head + loss fwd + bwd (torch)", head_only)
for p in net.trainable().values():
    p.grad = torch.zeros_like(p)
timed("optimizer step (foreach)", opt_only)
