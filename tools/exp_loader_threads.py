"""How fast the library's file pool (cf_load_npy_int16) fills a pinned batch, by thread count: 1110 x 4096-sample int16 .npy files
(one CLI batch, page cache warm), then `cli.run_pipeline` over 12 500 such files per precision -- where the loader, not the GPU, sets
the pace (bf16), the thread count is the lever.   usage: python tools/exp_loader_threads.py"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from catfish_amd import _native as N, checkpoint, cli, placement  # noqa: E402

placement.bind(0, 1)
root = tempfile.mkdtemp(prefix="catfish_loader_")
try:
    reads = os.path.join(root, "reads")
    os.makedirs(reads)
    rng = np.random.default_rng(0)
    n_files = 12500
    for i in range(n_files):
        np.save(os.path.join(reads, "read_%06d.npy" % i), bench.squiggle_dac(rng, 4096))
    names = sorted(os.listdir(reads))
    paths = [os.fsencode(os.path.join(reads, n)) for n in names[:1110]]
    blob = b"\x00".join(paths) + b"\x00"
    bounds = np.zeros(len(paths) + 1, dtype=np.int64)
    np.cumsum([len(p) + 1 for p in paths], out=bounds[1:])
    stage = torch.empty(1110 * 4096, dtype=torch.int16, pin_memory=True)
    lengths = np.empty(len(paths), dtype=np.int64)
    total = C.c_int64(0)
    lib = N.lib()
    for nt in (1, 2, 4, 8, 16, 32):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            rc = lib.cf_load_npy_int16(blob, bounds.ctypes.data_as(C.c_void_p), len(paths), C.c_void_p(stage.data_ptr()), stage.numel(),
                                       lengths.ctypes.data_as(C.c_void_p), C.byref(total), nt)
            best = min(best, time.perf_counter() - t0)
        assert rc == 0
        print("loader %2d threads: %.2f ms per 1110-file batch = %.0f M samples/s" % (nt, best * 1e3, 1110 * 4096 / best / 1e6), flush=True)
    os.makedirs(os.path.join(root, "ResNetRNN", "checkpoints"))
    with open(os.path.join(root, "ResNetRNN", "ResNetRNN.txt"), "w") as fh:
        fh.write("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\nlearning_rate: 0.001\nlayer_size: 64\nn_layers: 3\n"
                 "keep_prob: 0.8\nlayer_size_res: 32\nn_layers_res: 2\n")
    checkpoint.write_checkpoint(os.path.join(root, "ResNetRNN", "checkpoints", "ckpnt-30000"), bench.load_weights())
    import contextlib
    import io
    for prec in ("fp32", "bf16x3", "bf16"):
        for rep in range(2):
            out = os.path.join(root, "out_%s_%d" % (prec, rep))
            timings = {}
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                cli.run_pipeline(reads, out, network_path=os.path.join(root, "ResNetRNN"), device=0, precision=prec, timings=timings)
            dt = time.perf_counter() - t0 - timings["model_s"]
            print("cli %-6s run %d: %.1f M samples/s  (%.4f s: listing %.4f infer %.4f chunks %.4f write %.4f)" % (
                prec, rep, n_files * 4096 / dt / 1e6, dt, timings["listing_s"], timings["infer_s"], timings["chunks_s"], timings["write_s"]), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
