"""Per-kernel times of the any-size path (profile slots of the engine) at a few geometries."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from catfish_amd.engine import HipEngine
from oracle import catfish_oracle as oracle


def main():
    n = 256 * 118
    x = torch.randn(n, 35, device="cuda")
    out = torch.empty(n * 35, device="cuda")
    geoms = ((64, 32, 3, 2), (128, 64, 3, 2), (256, 128, 2, 2), (32, 64, 3, 2))
    if os.environ.get("GEN"):                       # one geometry "h,c,layers,blocks" (tools/pmc_generic.sh)
        geoms = (tuple(int(v) for v in os.environ["GEN"].split(",")),)
    for h, c, nl, nr in geoms:
        w = oracle.random_weights(seed=1, layer_size=h, n_layers=nl, layer_size_res=c, n_layers_res=nr)
        os.environ["CATFISH_DEBUG_KNOBS"] = os.environ["CATFISH_GENERIC"] = "1"
        eng = HipEngine(w, layer_size=h, n_layers=nl, layer_size_res=c, n_layers_res=nr, device=0, max_windows_per_pass=n)
        os.environ.pop("CATFISH_GENERIC", None)
        for _ in range(2):
            eng.infer_device(x, out=out)
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(3):
            eng.infer_device(x, out=out)
        torch.cuda.synchronize()
        k = eng.profile_read()
        eng.profile_enable(False)
        print(json.dumps(dict(layer_size=h, layer_size_res=c, n_layers=nl, n_layers_res=nr,
                              ms_per_pass={name: v[0] / 3 for name, v in k.items()})), flush=True)
        eng.close()


main()
