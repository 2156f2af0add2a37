"""CPU baseline leg of bench.py -- TEST/BENCH INFRASTRUCTURE, never on the product path.

Times the fp32 numpy oracle (restatement of the reference's TF-1.10 CPU graph) with the
reference's call pattern: one forward per read, batch = that read's windows
(catfish/infer.py:44).  Reads are independent, so the fairest use of the host is one
single-threaded worker process per core.
"""
from __future__ import annotations

import os
import time

import numpy as np


def _worker(args):
    seed, read_len, budget_s, weights_path = args
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    from oracle import catfish_oracle as oracle
    with np.load(weights_path) as z:
        weights = {k: z[k] for k in z.files}
    dac = oracle.synthetic_dac(8, read_len, seed=seed)
    xs = [oracle.pad_and_window(oracle.normalize_raw_signal(d))[0].astype(np.float32) for d in dac]
    oracle.forward(xs[0], weights, np.float32)  # warm-up (page in BLAS)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        oracle.forward(xs[n % len(xs)], weights, np.float32)
        n += 1
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return n, time.perf_counter() - t0


def affinity_cores():
    """Cores this process may run on (the GPU box reports 256)."""
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return ncpu


def run(weights_path, read_len=4096, budget_s=12.0, workers=None):
    """-> dict for bench.py's "cpu_baseline" object.  ``workers``: single-threaded worker processes (default 16 = a
    1-GPU box's CPU share), never more than the affinity mask allows."""
    import multiprocessing as mp
    ncpu = affinity_cores()
    workers = max(1, min(workers or 16, ncpu))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_worker, [(100 + i, read_len, budget_s, weights_path) for i in range(workers)])
    wall = time.perf_counter() - t0
    rate = sum(n * read_len / dt for n, dt in res)
    n_reads = sum(n for n, _ in res)
    return {"value": rate, "unit": "samples/s", "cores": workers, "kind": "port",
            "sample": "%d synthetic %d-sample reads in %.1f s per worker: %d single-threaded worker processes, "
                      "one fp32 numpy forward per read (restatement of the TF-1.10 CPU path, not TF itself); "
                      "host has %d cpus, pool wall %.1f s" % (n_reads, read_len, budget_s, workers, os.cpu_count() or 0, wall)}
