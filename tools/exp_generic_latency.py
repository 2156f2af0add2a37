"""Latency of small calls on the any-size path (one 4096-sample read = 118 windows, 1024 windows)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from catfish_amd.engine import HipEngine
from oracle import catfish_oracle as oracle

for h, c in ((32, 16), (64, 32), (128, 64), (256, 128)):
    w = oracle.random_weights(seed=1, layer_size=h, layer_size_res=c)
    os.environ["CATFISH_DEBUG_KNOBS"] = os.environ["CATFISH_GENERIC"] = "1"
    eng = HipEngine(w, layer_size=h, n_layers=3, layer_size_res=c, n_layers_res=2, device=0, max_windows_per_pass=4096)
    os.environ.pop("CATFISH_GENERIC", None)
    res = dict(layer_size=h, layer_size_res=c)
    for n in (118, 1024, 4096):
        x = torch.randn(n, 35, device="cuda")
        out = torch.empty(n * 35, device="cuda")
        for _ in range(3):
            eng.infer_device(x, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.infer_device(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        res["ms_%d_windows" % n] = e0.elapsed_time(e1) / 10
    print(json.dumps(res), flush=True)
    eng.close()
