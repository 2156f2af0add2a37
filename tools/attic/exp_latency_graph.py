"""Does replaying a small call as a HIP graph shorten it?  One read (118 windows, fp32), synchronised after every call:
eager launches against a torch.cuda.CUDAGraph captured around the same cf_infer call (same buffers).
usage: python tools/exp_latency_graph.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(bench.load_weights(), device=0, max_windows_per_pass=4096)
for n in (118, 768):
    x = torch.randn(n, 35, device="cuda")
    y = torch.empty(n * 35, device="cuda")
    for _ in range(20):
        eng.infer_device(x, out=y)
    torch.cuda.synchronize()
    ref = y.clone()

    def timed(fn, reps=300):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    eager = timed(lambda: eng.infer_device(x, out=y))
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eng.infer_device(x, out=y, stream=side)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            eng.infer_device(x, out=y, stream=side)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref), "graph replay differs"
    graph = timed(g.replay)
    print("%d windows: eager %.4f ms per call, graph replay %.4f ms" % (n, eager, graph))
eng.close()
