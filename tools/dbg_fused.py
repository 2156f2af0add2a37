import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from catfish_amd.engine import HipEngine
import bench
w = bench.load_weights()
for n in (16, 129, 1000, 30208):
    for fuse in (False, True):
        eng = HipEngine(w, device=0, max_windows_per_pass=32768, fuse_layers=fuse)
        x = np.random.default_rng(0).normal(size=(n, 35)).astype(np.float32)
        t0 = time.perf_counter()
        try:
            y = eng.infer_host(x)
            dt = time.perf_counter() - t0
            print(n, fuse, "%.3f s" % dt, float(y[:5].sum()), flush=True)
        except Exception as e:
            print(n, fuse, "ERR", e, "%.3f s" % (time.perf_counter() - t0), flush=True)
        eng.close()
