"""End-to-end rate of the file-driven path behind the CLI (catfish/catfish:50-56): a directory of .npy reads -> infer_files_sharded
(load, device normalisation, forward pass, post-processing, spans) on one rank."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from catfish_amd import sharding  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    d = tempfile.mkdtemp(dir="/tmp")
    _, dacs = bench.make_reads(256, seed=9, return_dac=True)
    paths = []
    for i in range(n):
        p = os.path.join(d, "r%06d.npy" % i)
        np.save(p, dacs[i % 256])
        paths.append(p)
    from catfish_amd.resnet_class import ResNetRNN
    model = ResNetRNN(batch_size=256, optimizer_choice="RMSProp", learning_rate=0.001, layer_size=64, n_layers=3, keep_prob=0.8,
                      layer_size_res=32, n_layers_res=2)
    model.set_weights(bench.load_weights())
    sharding.infer_files_sharded(model, paths[:512], rank=0, world_size=1)          # warm-up
    for label in ("first", "second"):
        t0 = time.perf_counter()
        res = sharding.infer_files_sharded(model, paths, rank=0, world_size=1)
        dt = time.perf_counter() - t0
        print(json.dumps(dict(run=label, files=n, seconds=dt, samples_per_s=n * 4096 / dt, files_per_s=n / dt,
                              spans=sum(len(r[0]) for r in res))), flush=True)
    t0 = time.perf_counter()
    from catfish_amd import infer
    for p in paths:
        infer.load_dac(p)
    print(json.dumps(dict(load_only_seconds=time.perf_counter() - t0)))
    shutil.rmtree(d)


main()
