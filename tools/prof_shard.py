import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from catfish_amd.engine import HipEngine
from catfish_amd.pipeline import ReadPipeline
from catfish_amd.batching import spans_from_runs
w = bench.load_weights()
eng = HipEngine(w, device=0, max_windows_per_pass=256*118)
rng = np.random.default_rng(1)
base = [bench.squiggle_dac(rng, 4096) for _ in range(256)]
pipe = ReadPipeline(eng, 256*4096)
junk = None
def busy(ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms: pass
import gc
for mode in ("arrays", "lists", "nogc:lists"):
    gc.enable(); gc.unfreeze()
    if mode.startswith("nogc"): gc.disable()
    if mode.startswith("freeze"): gc.collect(); gc.freeze()
    for _ in pipe.run([base]*4, as_lists=False): pass
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    pending = None
    for k in range(24):
        a = time.perf_counter(); tk = pipe.submit(base); b = time.perf_counter()
        if pending is not None:
            pending.done.synchronize(); c = time.perf_counter()
            r = pipe.collect(pending, mode.endswith("lists"))
            if mode == "arrays+sleep2ms": time.sleep(0.002)
            if mode == "arrays+spin2ms": busy(2.0)
            if mode == "arrays+tolist":
                junk = np.stack([r[1], r[2]], 1).tolist()
            if mode == "arrays+stack":
                for _ in range(20): junk = np.stack([r[1], r[2]], 1)
            if mode == "arrays+tolist1d":
                junk = (r[1].tolist(), r[2].tolist())
            if mode == "arrays+bytearray":
                for _ in range(50): junk = bytearray(1 << 20)
            if mode.endswith("arrays+pylist"):
                junk = [[i, i + 1] for i in range(10000)]
            d = time.perf_counter()
            ev.append((b - a, c - b, d - c))
        pending = tk
    pipe.collect(pending, False)
    tot = time.perf_counter() - t0
    s = np.array(ev) * 1e3
    print("%-18s %.2f ms per batch; submit %.2f  wait %.2f  collect %.2f ms (means)" % (mode, tot / 24 * 1e3, *s.mean(0)))
