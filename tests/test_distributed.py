"""N > 1 path on CPU: two gloo ranks shard reads, run their shard, host-gather on rank 0.

The per-shard inference callable is injected; here it is backed by the CPU oracle (tests may use
the oracle as a stand-in engine -- the product path passes the HIP engine).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from catfish_amd import sharding, infer
    from oracle import catfish_oracle as oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = oracle.random_weights(seed=3)
        rng = np.random.default_rng(11)
        lens = [36, 700, 35, 140, 999, 70, 512, 64, 300]
        sigs = [rng.normal(size=n) for n in lens]

        def infer_fn(batch):
            out = []
            for s in batch:
                spans, n, _ = oracle.infer_read(s, w, np.float32)
                out.append((spans, n))
            return out

        res = sharding.run_sharded(sigs, infer_fn, gather_group=sharding.host_gather_group())
        if rank == 0:
            want = infer_fn(sigs)
            assert res == want
            assert [r[1] for r in res] == lens
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_single_process_run_sharded_without_process_group():
    sys.path.insert(0, ROOT)
    from catfish_amd import sharding
    sigs = [np.zeros(n) for n in (10, 20, 30)]
    res = sharding.run_sharded(sigs, lambda b: [len(s) for s in b])
    assert res == [10, 20, 30]
