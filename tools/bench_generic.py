"""Throughput of the any-size inference path (csrc/generic.hpp) at a few geometries of the reference's hyper-parameter search,
next to the tuned kernels at the shipped one: samples/s and the fraction of the fp32-MFMA peak (flops from the layer sizes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from catfish_amd.engine import HipEngine
from oracle import catfish_oracle as oracle

PEAK = 157.3e12


def flops_per_sample(h, c, n_layers, n_layers_res):
    mac = 0
    cin = 1
    for _ in range(n_layers_res):
        mac += 2 * cin * c + 3 * c * c + c * c
        cin = c
    cin = c if n_layers_res else 1
    for _ in range(n_layers):
        mac += 2 * 3 * h * (cin + h)
        cin = 2 * h
    return 2 * (mac + 2 * h)


def main():
    n = 256 * 118
    x = torch.randn(n, 35, device="cuda")
    out = torch.empty(n * 35, device="cuda")
    for h, c, nl, nr, force in ((64, 32, 3, 2, False), (64, 32, 3, 2, True), (64, 64, 3, 2, False), (64, 128, 3, 2, False), (16, 16, 2, 2, False), (32, 64, 3, 2, False),
                                (128, 64, 3, 2, False), (128, 128, 2, 4, False), (256, 128, 3, 2, False), (256, 256, 5, 5, False)):
        w = oracle.random_weights(seed=1, layer_size=h, n_layers=nl, layer_size_res=c, n_layers_res=nr)
        if force:
            os.environ["CATFISH_DEBUG_KNOBS"] = os.environ["CATFISH_GENERIC"] = "1"
        eng = HipEngine(w, layer_size=h, n_layers=nl, layer_size_res=c, n_layers_res=nr, device=0, max_windows_per_pass=n)
        os.environ.pop("CATFISH_GENERIC", None)
        reps = 3 if h >= 128 else 10
        for _ in range(2):
            eng.infer_device(x, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.infer_device(x, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        sps = 256 * 4096 / (ms * 1e-3)
        fl = flops_per_sample(h, c, nl, nr)
        print(json.dumps(dict(layer_size=h, layer_size_res=c, n_layers=nl, n_layers_res=nr,
                              path="any-size" if eng.launch_regimes()["coop_max"] == 0 else "tuned", ms_per_256_reads=ms,
                              samples_per_s=sps, flop_per_sample=fl, frac_of_fp32_mfma_peak=sps * fl / PEAK)), flush=True)
        eng.close()


main()
