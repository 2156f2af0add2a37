"""Sweep the chunking of the fused bf16 residual stack (res_stack2_bf16_kernel): positions per wave task and tiles per wave.
usage: CATFISH_DEBUG_KNOBS=1 python tools/exp_res_bf16.py [n_windows]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CATFISH_DEBUG_KNOBS"] = "1"
import torch
from catfish_amd.engine import HipEngine
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30208
w = bench.load_weights()
x = torch.randn(n, 35, device="cuda")
y = torch.empty(n * 35, device="cuda")
out = {}
for prec in ("bf16", "bf16x3"):
    eng = HipEngine(w, device=0, max_windows_per_pass=max(n, 4096), precision=prec)
    for tpw in ((1, 2) if prec == "bf16" else (1,)):
        os.environ["CATFISH_RES_TPW"] = str(tpw)
        for chunks in (0, 1, 2, 3, 4, 5, 7, 9, 12, 18, 35):
            if chunks:
                os.environ["CATFISH_RES_CHUNKS"] = str(chunks)
            else:
                os.environ.pop("CATFISH_RES_CHUNKS", None)
            for _ in range(5):
                eng.infer_device(x, out=y)
            torch.cuda.synchronize()
            eng.profile_enable(True, every=1)
            eng.profile_reset()
            for _ in range(20):
                eng.infer_device(x, out=y)
            k = eng.profile_read()
            eng.profile_enable(False)
            out["%s tpw%d chunks%s" % (prec, tpw, chunks or "auto")] = round(k["res_stack2"][0] / k["res_stack2"][1] * 1e3, 1)
    os.environ.pop("CATFISH_RES_CHUNKS", None)
    os.environ["CATFISH_RES_FUSE"] = "0"
    eng.profile_enable(True, every=1)
    eng.profile_reset()
    for _ in range(20):
        eng.infer_device(x, out=y)
    k = eng.profile_read()
    out["%s two launches" % prec] = round(sum(k[s][0] / k[s][1] for s in ("res_block_first", "res_block")) * 1e3, 1)
    os.environ.pop("CATFISH_RES_FUSE")
    eng.close()
print(json.dumps({"n_windows": n, "res_stack2_us": out}, indent=1))
