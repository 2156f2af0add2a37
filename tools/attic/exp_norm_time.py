import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from catfish_amd.engine import HipEngine
from catfish_amd import batching
w = bench.load_weights()
eng = HipEngine(w, device=0, max_windows_per_pass=256*118, precision="bf16")
rng = np.random.default_rng(1)
base = [bench.squiggle_dac(rng, 4096) for _ in range(256)]
for _ in range(3): batching.infer_reads_dac(eng, base, max_windows=256*118)
eng.profile_enable(True, every=1); eng.profile_reset()
for _ in range(10): batching.infer_reads_dac(eng, base, max_windows=256*118)
k = eng.profile_read()
print({n: round(v[0]/v[1]*1e3,1) for n,v in k.items()})
