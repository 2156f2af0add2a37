"""N > 1 path on CPU: two gloo ranks shard reads, run their shard, host-gather on rank 0.

The per-shard inference callable is injected; here it is backed by the CPU oracle (tests may use
the oracle as a stand-in engine -- the product path passes the HIP engine).
"""
import json
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


class _OracleBatchRunner(object):
    """Stand-in for sharding.EngineBatchRunner on a box without a GPU: the same interface (an iterable of raw-read
    batches in, per-batch [(spans, length)] out), backed by the CPU oracle."""

    def __init__(self, weights):
        self.w = weights
        self.batches = []

    def run(self, batches):
        from oracle import catfish_oracle as oracle
        for reads in batches:
            self.batches.append([len(r) for r in reads])
            out = []
            for r in reads:
                spans, n, _ = oracle.infer_read(oracle.normalize_raw_signal(np.asarray(r)), self.w, np.float32)
                out.append((spans, n))
            yield out


def _runner_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from catfish_amd import sharding
    from oracle import catfish_oracle as oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    assert sharding.dist_env() == (rank, world, rank)
    assert sharding.init_host_group()                  # the product's own group set-up (gloo)
    try:
        w = oracle.random_weights(seed=3)
        lens = [36, 700, 35, 140, 999, 70, 512, 64, 300, 2000, 1]
        dacs = [oracle.synthetic_dac(1, max(n, 2), seed=40 + i)[0][:n] for i, n in enumerate(lens)]
        paths = []
        for i, d in enumerate(dacs):
            p = os.path.join(tmpdir, "read%02d.npy" % i)
            if rank == 0:
                np.save(p, d)
            paths.append(p)
        dist.barrier()
        fake = _OracleBatchRunner(w)
        # in-memory reads, batches bounded by samples
        res = sharding.infer_reads_sharded(None, dacs, max_samples_per_batch=1200, batch_runner=fake,
                                           gather_group=sharding.host_gather_group())
        mine = sharding.shard_reads(lens, world)[rank]
        assert sorted(sum(fake.batches, [])) == sorted(lens[i] for i in mine)      # this rank ran its own shard only
        assert all(sum(b) <= 1200 or len(b) == 1 for b in fake.batches)
        # files: every rank loads only its own shard
        fake2 = _OracleBatchRunner(w)
        res_f = sharding.infer_files_sharded(None, paths, max_samples_per_batch=1200, batch_runner=fake2)
        assert 0 < len(sum(fake2.batches, [])) < len(lens)
        if rank == 0:
            want = [oracle.infer_read(oracle.normalize_raw_signal(d), w, np.float32)[:2] for d in dacs]
            want = [(s, n) for s, n in want]
            assert res == want and res_f == want
            open(os.path.join(tmpdir, "ok2"), "w").write("ok")
        else:
            assert res is None and res_f is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_runner_with_fake_engine(tmp_path):
    """The product's sharded runner (what catfish_amd.cli calls) on two gloo ranks; the per-rank engine is replaced
    by an oracle-backed batch runner because this box has no GPU (tests/test_gpu_pipeline.py drives the real one)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_runner_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok2").exists()


def test_span_table_round_trips_the_reference_result_lists():
    """What a rank sends through the host gather: four arrays that expand to exactly the reference's per-read
    ``(spans, length)`` (infer.py:12-51), empty reads and empty shards included."""
    sys.path.insert(0, ROOT)
    from catfish_amd.sharding import SpanTable
    rng = np.random.default_rng(0)
    res = []
    for _ in range(200):
        k = int(rng.integers(0, 6))
        starts = np.sort(rng.integers(-11, 4000, size=k))
        res.append(([[int(s), int(s) + 42] for s in starts], int(rng.integers(1, 5000))))
    t = SpanTable.from_lists(res)
    assert len(t) == 200 and t.expand() == res
    parts = [SpanTable.from_lists(res[:50]), SpanTable.from_lists([]), SpanTable.from_lists(res[50:51]), SpanTable.from_lists(res[51:])]
    assert SpanTable.concat(parts).expand() == res
    assert SpanTable.concat([]).expand() == [] and SpanTable.from_lists([([], 7)]).expand() == [([], 7)]
    import pickle
    assert pickle.loads(pickle.dumps(t)).expand() == res           # it travels through dist.gather_object as a pickle


def test_shard_costs_balances_and_covers():
    sys.path.insert(0, ROOT)
    from catfish_amd import sharding
    rng = np.random.default_rng(5)
    costs = rng.integers(1, 500, size=200)
    for world in (1, 2, 3, 8):
        shards = sharding.shard_costs(costs, world)
        assert sorted(sum(shards, [])) == list(range(200))
        loads = [int(costs[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= int(costs.max())          # LPT bound
    assert sharding.shard_costs([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]
    with pytest.raises(ValueError):
        sharding.shard_costs([1], 0)


def test_single_process_run_sharded_without_process_group():
    sys.path.insert(0, ROOT)
    from catfish_amd import sharding
    sigs = [np.zeros(n) for n in (10, 20, 30)]
    res = sharding.run_sharded(sigs, lambda b: [len(s) for s in b])
    assert res == [10, 20, 30]


def test_prefetched_loader_keeps_order_and_raises():
    """sharding._prefetched: batches come out in order from the loader thread; an exception inside the generator (a bad
    file) surfaces in the consumer."""
    from catfish_amd import sharding
    assert list(sharding._prefetched(iter(range(50)), depth=3)) == list(range(50))
    assert list(sharding._prefetched(iter(()), depth=2)) == []

    def bad():
        yield 1
        yield 2
        raise ValueError("path to FAST5 is not correct.")

    got = []
    with pytest.raises(ValueError, match="FAST5"):
        for v in sharding._prefetched(bad(), depth=2):
            got.append(v)
    assert got == [1, 2]


def _failing_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from catfish_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def work(mine):
            if rank == 1:
                raise ValueError("path to FAST5 is not correct.")          # what a missing file raises (infer.py:25-26)
            return [("ok", i) for i in mine]

        try:
            sharding.run_sharded_indexed([1] * 10, work)
            outcome = "returned"
        except ValueError as exc:
            outcome = "ValueError %s" % exc
        except RuntimeError as exc:
            outcome = "RuntimeError %s" % exc
        open(os.path.join(tmpdir, "rank%d" % rank), "w").write(outcome)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_a_failing_rank_fails_the_whole_sharded_run_without_hanging(tmp_path):
    """One rank's shard raises (a missing file): it still takes part in the gather, so nobody waits for it until the group
    times out; the failing rank re-raises its own error, the others raise a RuntimeError that names it."""
    import torch.multiprocessing as mp
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (tmp_path / "rank0").read_text(), (tmp_path / "rank1").read_text()
    assert r1.startswith("ValueError") and "FAST5" in r1
    assert r0.startswith("RuntimeError") and "rank 1" in r0 and "FAST5" in r0


def _eight_rank_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from catfish_amd import sharding
    from oracle import catfish_oracle as oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert sharding.init_host_group()
    try:
        w = oracle.random_weights(seed=3, layer_size=16, n_layers=1, layer_size_res=16, n_layers_res=1)
        w["final_fully_connected/bias"] = np.array([0.4], np.float32)          # random weights: some samples over 0.5

        def infer_read(r):
            x = oracle.pad_and_window(oracle.normalize_raw_signal(np.asarray(r)))[0]
            p = oracle.forward(x, w, np.float32, n_layers=1, n_layers_res=1)[:len(r)]
            return oracle.hp_in_pred(oracle.correct_short(oracle.class_from_threshold(p))), len(r)

        class Runner(sharding.EngineBatchRunner):
            def __init__(self):
                pass

            def run(self, batches, compact=False):
                for reads in batches:
                    res = [infer_read(r) for r in reads]
                    yield sharding.SpanTable.from_lists(res) if compact else res

        for n_reads in (19, 5):                        # more reads than ranks; fewer reads than ranks (empty shards)
            dacs = [oracle.synthetic_dac(1, 300 + 97 * i, seed=70 + i)[0] for i in range(n_reads)]
            res = sharding.infer_reads_sharded(None, dacs, max_samples_per_batch=900, batch_runner=Runner(),
                                               gather_group=sharding.host_gather_group())
            flat = sharding.infer_reads_sharded(None, dacs, max_samples_per_batch=900, batch_runner=Runner(),
                                                gather_group=sharding.host_gather_group(), as_table=True)
            if rank == 0:
                want = [infer_read(d) for d in dacs]
                assert res == want
                assert isinstance(flat, sharding.SpanTable) and flat.expand() == want      # contiguous blocks, one flat table
                assert [flat.read(i) for i in range(n_reads)] == want
                open(os.path.join(tmpdir, "ok8_%d" % n_reads), "w").write("ok")
            else:
                assert res is None and flat is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_gather_of_span_tables_with_empty_shards(tmp_path):
    """The N = 8 shape of BASELINE configs[2] on CPU (gloo): eight ranks, LPT shards, each rank hands a SpanTable to the host
    gather, rank 0 rebuilds the per-read lists in input order -- with more reads than ranks and with fewer (three ranks hold
    nothing and still take part in the collective)."""
    import torch.multiprocessing as mp
    mp.spawn(_eight_rank_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert (tmp_path / "ok8_19").exists() and (tmp_path / "ok8_5").exists()


# ------------------------------------------------------------------ the CLI with its tail on every rank (catfish/catfish:50-82)
def _small_oracle_runner():
    """An EngineBatchRunner whose device work is the CPU oracle on a tiny random network (16 units, one layer)."""
    from catfish_amd import sharding
    from oracle import catfish_oracle as oracle
    w = oracle.random_weights(seed=3, layer_size=16, n_layers=1, layer_size_res=16, n_layers_res=1)
    w["final_fully_connected/bias"] = np.array([0.4], np.float32)          # random weights: some samples over 0.5

    def infer_read(r):
        x = oracle.pad_and_window(oracle.normalize_raw_signal(np.asarray(r)))[0]
        p = oracle.forward(x, w, np.float32, n_layers=1, n_layers_res=1)[:len(r)]
        return oracle.hp_in_pred(oracle.correct_short(oracle.class_from_threshold(p))), len(r)

    class Runner(sharding.EngineBatchRunner):
        def __init__(self, model=None, max_samples_per_batch=None):
            self.loaded = 0

        def run(self, batches, compact=False):
            for reads in batches:
                self.loaded += len(reads)
                res = [infer_read(r) for r in reads]
                yield sharding.SpanTable.from_lists(res) if compact else res

        def run_files(self, path_batches, compact=False):          # the product reads the files natively into pinned memory
            from catfish_amd.infer import load_dac
            return self.run(([load_dac(p) for p in paths] for paths in path_batches), compact)

    return Runner, infer_read


def _write_reads(directory, n_reads):
    from oracle import catfish_oracle as oracle
    os.makedirs(directory, exist_ok=True)
    for i in range(n_reads):
        np.save(os.path.join(directory, "read_%03d.npy" % i), oracle.synthetic_dac(1, 900 + 211 * (i % 7), seed=300 + i)[0])


def _pipeline_worker(rank, world, port, tmpdir, n_reads, out_name, break_setup):
    sys.path.insert(0, ROOT)
    import time
    from catfish_amd import cli, sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CATFISH_DIST_TIMEOUT_S="60")
    Runner, _ = _small_oracle_runner()
    sharding.EngineBatchRunner = Runner                                    # no GPU here: the oracle stands in for the engine

    def load_network(*a, **k):
        if break_setup == "network" and rank == world - 1:
            raise ValueError("checkpoint is missing tensor 'conv1d/kernel'")
        return object()

    cli.neural_network.load_network = load_network
    if break_setup == "write" and rank == world - 1:                      # this rank's disk is full when it writes its members
        import errno
        from catfish_amd import chunks

        def no_space(fd, data, offset):
            raise OSError(errno.ENOSPC, "No space left on device")
        chunks._pwrite_all = no_space
    if break_setup == "split" and rank == world - 1:                      # ... or when it writes its chunk files
        import errno
        from catfish_amd import split

        def no_space_for_chunks(table, listing, lo, hp_dir, nonhp_dir, n_threads=None):
            raise OSError(errno.ENOSPC, "No space left on device", hp_dir)
        split.split_listing = no_space_for_chunks                          # (what cf_listing_split_npy_int16's CF_ERR_IO becomes)
    if break_setup == "short":                                             # every pwrite stops after 7 bytes
        real_pwrite = os.pwrite
        os.pwrite = lambda fd, data, offset: real_pwrite(fd, bytes(data[:7]), offset)
    if break_setup == "write-type" and rank == world - 1:                 # not an OSError: the agreement must still be reached
        from catfish_amd import chunks

        def wrong_type(fd, data, offset):
            raise TypeError("a bytes-like object is required, not 'str'")
        chunks._pwrite_all = wrong_type
    if break_setup == "listing" and rank == world - 1:                     # this rank does not see the last file (yet): to it the
        real_from = sharding.DirListing.from_names_blob                    # input directory is a copy without that file
        sharding.DirListing.from_names_blob = classmethod(
            lambda cls, directory, blob, n: real_from(os.path.join(tmpdir, "reads_stale"), blob, n))
    if break_setup == "count-stats":                                       # every entry this rank asks the sizes of is counted
        real_sizes, seen = sharding.DirListing.sizes, []

        def counting(self, lo, hi, *a, **k):
            seen.extend(range(lo, hi))
            return real_sizes(self, lo, hi, *a, **k)
        sharding.DirListing.sizes = counting
    t0 = time.time()
    try:
        res = cli.run_pipeline(os.path.join(tmpdir, "reads"), os.path.join(tmpdir, out_name), chunk_size=300,
                               gather_table=(out_name == "out8"))
        table = res.get("table")
        outcome = "returned %d reads, table %s" % (res["reads"], None if table is None else len(table))
    except Exception as exc:                                               # noqa: BLE001 -- recorded for the parent
        outcome = "%s %s" % (type(exc).__name__, exc)
    if break_setup == "count-stats":
        outcome += " stats %d" % len(seen)
    open(os.path.join(tmpdir, "%s.rank%d" % (out_name, rank)), "w").write("%.1f %s" % (time.time() - t0, outcome))


@pytest.mark.timeout(300)
def test_eight_ranks_write_the_same_bytes_as_one(tmp_path):
    """catfish -i -s over 8 gloo ranks vs 1: every rank classifies AND merges its own reads (native chunk tables) and writes
    its own part of the two documents at its offset -- byte-identical documents, equal to the per-read Python rules on the
    per-read results; some ranks hold no read with a homopolymer chunk, or no read at all."""
    import json
    import torch.multiprocessing as mp
    from catfish_amd import cli
    n_reads = 21
    _write_reads(str(tmp_path / "reads"), n_reads)
    mp.spawn(_pipeline_worker, args=(8, _free_port(), str(tmp_path), n_reads, "out8", None), nprocs=8, join=True)
    mp.spawn(_pipeline_worker, args=(1, _free_port(), str(tmp_path), n_reads, "out1", None), nprocs=1, join=True)
    assert (tmp_path / "out8.rank0").read_text().endswith("returned %d reads, table %d" % (n_reads, n_reads))   # gather_table=True
    assert all((tmp_path / ("out8.rank%d" % r)).read_text().endswith("returned %d reads, table None" % n_reads) for r in range(1, 8))
    assert (tmp_path / "out1.rank0").read_text().endswith("returned %d reads, table None" % n_reads)
    docs = {}
    for out in ("out8", "out1"):
        docs[out] = [(tmp_path / out / "TEMP" / f).read_bytes() for f in ("hp_positions.json", "nonhp_positions.json")]
    assert docs["out8"] == docs["out1"]
    hp, nonhp = (json.loads(d) for d in docs["out8"])
    _, infer_read = _small_oracle_runner()
    assert len(nonhp) == n_reads and 0 < len(hp) <= n_reads
    for name in sorted(os.listdir(tmp_path / "reads")):
        spans, length = infer_read(np.load(tmp_path / "reads" / name))
        merged, non = cli.chunks_of_read([list(s) for s in spans], length, 300)
        assert hp.get(name) == merged and nonhp[name] == json.loads(json.dumps(non))
    # the split step (catfish/catfish:85-92 -> split_f5.py:8-81): every rank cut ITS reads -- same file set and same bytes as one
    # rank, each file numpy's save of signal[s0:s1], the index running on from the HP chunks into the non-HP ones
    import io
    expected = {}
    for name in hp:
        signal = np.load(tmp_path / "reads" / name)
        for k, (s0, s1) in enumerate(hp[name] + nonhp[name]):
            buf = io.BytesIO()
            np.save(buf, signal[s0:s1])
            expected["%s/%s_%d.npy" % ("HP" if k < len(hp[name]) else "nonHP", name.split(".")[0], k)] = buf.getvalue()
    for out in ("out8", "out1"):
        got = {"%s/%s" % (d, f): (tmp_path / out / "TEMP" / d / f).read_bytes()
               for d in ("HP", "nonHP") for f in os.listdir(tmp_path / out / "TEMP" / d)}
        assert got == expected and len(got) > len(hp)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("what", ["directories", "network", "file"])
def test_a_rank_that_fails_during_set_up_fails_the_job_at_once(tmp_path, what):
    """a failure BEFORE the data path (rank 0: the split directory exists, catfish/catfish:37-38; any rank:
    the network does not load) used to leave the failing rank in a barrier and the others in the gather until the group
    timed out.  Now the ranks compare notes after set-up: the failing one raises its own error, the others name it."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 4)
    if what == "directories":
        os.makedirs(tmp_path / "out" / "TEMP" / "HP")
    if what == "file":                                   # the last file (rank 1's block) is not a read
        (tmp_path / "reads" / "read_zzz.npy").write_bytes(b"not a numpy file")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 4, "out", what), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    if what == "directories":
        assert r0[1].startswith("FileExistsError") and r1[1].startswith("RuntimeError") and "rank 0: FileExistsError" in r1[1]
    elif what == "network":
        assert r1[1].startswith("ValueError") and r0[1].startswith("RuntimeError") and "rank 1: ValueError" in r0[1]
    else:
        assert not r1[1].startswith("returned") and r0[1].startswith("RuntimeError") and "classification failed on rank 1" in r0[1]
        assert not (tmp_path / "out" / "TEMP" / "hp_positions.json").exists()      # nothing half-written


@pytest.mark.timeout(120)
def test_a_rank_whose_write_fails_fails_the_job_at_once(tmp_path):
    """VERDICT r03 / ADVICE r03: writing the documents was the one stage of the sharded CLI whose failure the ranks did not
    agree on -- an I/O error on one rank left its peers in a barrier until the group timed out (1800 s by default) and the
    pre-sized documents under their final names with NUL holes.  Now: the failing rank raises its own OSError, the others
    name it, within seconds, and neither a document nor a .part is left (an exception aborts the run: split_f5.py:23-32)."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 6)
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 6, "out", "write"), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    assert r1[1].startswith("OSError") and "No space left" in r1[1]
    assert r0[1].startswith("RuntimeError") and "writing the chunk documents failed on rank 1: OSError" in r0[1]
    assert sorted(os.listdir(tmp_path / "out" / "TEMP")) == ["HP", "nonHP"]         # nothing half-written, no .part left


@pytest.mark.timeout(120)
def test_a_rank_whose_split_fails_fails_the_job_at_once(tmp_path):
    """The split step is per rank and ends in an agreement: one rank's full disk raises on every rank within seconds (an exception
    aborts the run, split_f5.py:23-32); the documents, agreed on before, stay."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 6)
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 6, "out", "split"), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    assert r1[1].startswith("OSError") and "No space left" in r1[1]
    assert r0[1].startswith("RuntimeError") and "splitting the reads failed on rank 1: OSError" in r0[1]
    assert (tmp_path / "out" / "TEMP" / "hp_positions.json").exists()


@pytest.mark.timeout(120)
def test_a_rank_whose_write_fails_with_any_exception_fails_the_job_at_once(tmp_path):
    """``stage()`` caught only OSError, so a TypeError / MemoryError out of one rank's write skipped the agreement on
    that rank and left its peer in the all-gather until the group timed out.  Every exception now goes to the agreement."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 6)
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 6, "out", "write-type"), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    assert r1[1].startswith("TypeError") and "bytes-like" in r1[1]
    assert r0[1].startswith("RuntimeError") and "writing the chunk documents failed on rank 1: TypeError" in r0[1]
    assert sorted(os.listdir(tmp_path / "out" / "TEMP")) == ["HP", "nonHP"]


@pytest.mark.timeout(120)
def test_every_rank_stats_only_its_block_of_the_listing(tmp_path):
    """the listing of the input directory (catfish/catfish:49-50) used to be a scandir + stat of EVERY file on
    EVERY rank (800 000 stats for 100 000 files on 8 ranks) and was left out of the timed region.  Now rank 0 reads the names once
    and broadcasts them, every rank stats its n/world block, and the sizes travel in one all-gather.  4 ranks, 10 files:
    2 + 3 + 2 + 3 stats, same documents as 1."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 10)
    mp.spawn(_pipeline_worker, args=(4, _free_port(), str(tmp_path), 10, "out4", "count-stats"), nprocs=4, join=True)
    mp.spawn(_pipeline_worker, args=(1, _free_port(), str(tmp_path), 10, "out1", "count-stats"), nprocs=1, join=True)
    said = [(tmp_path / ("out4.rank%d" % r)).read_text() for r in range(4)]
    assert all("returned 10 reads" in t for t in said), said
    assert [int(t.rsplit(" ", 1)[1]) for t in said] == [2, 3, 2, 3]
    assert (tmp_path / "out1.rank0").read_text().endswith("stats 10")
    for f in ("hp_positions.json", "nonhp_positions.json"):
        assert (tmp_path / "out4" / "TEMP" / f).read_bytes() == (tmp_path / "out1" / "TEMP" / f).read_bytes()


@pytest.mark.timeout(120)
def test_short_writes_are_completed(tmp_path):
    """os.pwrite may write less than it was given; the documents must come out whole (2 ranks against 1, 7 bytes per call)."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 5)
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 5, "short", "short"), nprocs=2, join=True)
    mp.spawn(_pipeline_worker, args=(1, _free_port(), str(tmp_path), 5, "whole", None), nprocs=1, join=True)
    assert all((tmp_path / ("short.rank%d" % r)).read_text().endswith("returned 5 reads, table None") for r in (0, 1))
    for f in ("hp_positions.json", "nonhp_positions.json"):
        assert (tmp_path / "short" / "TEMP" / f).read_bytes() == (tmp_path / "whole" / "TEMP" / f).read_bytes()
        assert b"\0" not in (tmp_path / "short" / "TEMP" / f).read_bytes()
    assert not [f for f in os.listdir(tmp_path / "short" / "TEMP") if f.endswith(".part")]


@pytest.mark.timeout(120)
def test_a_rank_that_cannot_see_a_listed_file_fails_the_job_at_set_up(tmp_path):
    """ranks must not cut their blocks from different views of the input directory (a file still being copied in,
    stale NFS attributes): they would silently duplicate or drop reads.  Since round 5 rank 0 reads the directory once and
    broadcasts the ordered names (``sharding.shared_listing``), so there is ONE list; a rank from whose side a listed file is not
    there finds out when it stats its block, and every rank fails within seconds, naming rank and file."""
    import shutil
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 6)
    shutil.copytree(tmp_path / "reads", tmp_path / "reads_stale")
    gone = sorted(os.listdir(tmp_path / "reads_stale"))[-1]
    os.unlink(tmp_path / "reads_stale" / gone)
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 6, "out", "listing"), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    assert r1[1].startswith("ValueError") and gone in r1[1]
    assert r0[1].startswith("RuntimeError") and "listing the input directory failed on rank 1" in r0[1] and gone in r0[1]
    assert not (tmp_path / "out" / "TEMP" / "hp_positions.json").exists()


@pytest.mark.timeout(120)
def test_an_unreadable_input_directory_fails_every_rank_with_rank_0s_error(tmp_path):
    """Rank 0 is the only reader of the directory: when it cannot (here: the path is a file), the broadcast carries its error."""
    import torch.multiprocessing as mp
    (tmp_path / "reads").write_text("not a directory")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path), 0, "out", None), nprocs=2, join=True)
    r0, r1 = ((tmp_path / ("out.rank%d" % r)).read_text().split(" ", 1) for r in (0, 1))
    assert float(r0[0]) < 30 and float(r1[0]) < 30
    assert r0[1].startswith("ValueError") and "cannot open directory" in r0[1]
    assert r1[1].startswith("RuntimeError") and "listing the input directory failed on rank 0: ValueError" in r1[1]


def test_contiguous_shards_tile_in_order_and_balance():
    sys.path.insert(0, ROOT)
    from catfish_amd import sharding
    rng = np.random.default_rng(9)
    costs = rng.integers(1, 500, size=300)
    for world in (1, 2, 3, 8):
        shards = sharding.shard_contiguous(costs, world)
        assert sum(shards, []) == list(range(300))                          # blocks in order: gathered tables need no re-ordering
        loads = [int(costs[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= 2 * int(costs.max())
    assert sharding.shard_contiguous([1, 1], 4) == [[], [0], [], [1]] and sharding.shard_contiguous([], 3) == [[], [], []]
    with pytest.raises(ValueError):
        sharding.shard_contiguous([1], 0)


@pytest.mark.timeout(300)
def test_more_ranks_than_reads_go_through_every_stage(tmp_path):
    """Eight ranks, five reads: three ranks own no file at all and still take part in the listing, the agreement on the documents
    and the split step (an empty block: nothing to cut, no native call) -- the job ends on every rank with the same totals, and the
    output equals one rank's."""
    import torch.multiprocessing as mp
    _write_reads(str(tmp_path / "reads"), 5)
    mp.spawn(_pipeline_worker, args=(8, _free_port(), str(tmp_path), 5, "outmany", None), nprocs=8, join=True)
    mp.spawn(_pipeline_worker, args=(1, _free_port(), str(tmp_path), 5, "outone", None), nprocs=1, join=True)
    assert all((tmp_path / ("outmany.rank%d" % r)).read_text().endswith("returned 5 reads, table None") for r in range(8))
    for rel in ("hp_positions.json", "nonhp_positions.json"):
        assert (tmp_path / "outmany" / "TEMP" / rel).read_bytes() == (tmp_path / "outone" / "TEMP" / rel).read_bytes()
    for d in ("HP", "nonHP"):
        many = {f: (tmp_path / "outmany" / "TEMP" / d / f).read_bytes() for f in os.listdir(tmp_path / "outmany" / "TEMP" / d)}
        one = {f: (tmp_path / "outone" / "TEMP" / d / f).read_bytes() for f in os.listdir(tmp_path / "outone" / "TEMP" / d)}
        assert many == one and many


# ------------------------------------------------------------------ bench.py: RCCL is a bounded probe beside the host group
def _probe_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo")
    t0 = time.time()
    res = bench.probe_rccl(dist, torch, 0, timeout_s=20)
    dist.barrier()                                                         # the host group still works after a failed probe
    one = torch.ones(1)
    dist.all_reduce(one)
    os.environ["CATFISH_RCCL_PROBE"] = "0"
    skipped = bench.probe_rccl(dist, torch, 0)
    open(os.path.join(tmpdir, "probe.rank%d" % rank), "w").write(json.dumps({"res": res, "seconds": time.time() - t0, "sum": float(one.item()),
                                                                              "skipped": skipped}))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_rccl_probe_failure_costs_a_note_not_the_run(tmp_path):
    """the timing barrier of bench.py is on the host (gloo) group whatever RCCL does, and RCCL is brought up as a
    PROBE over a second group with a short timeout.  Here (no GPU: the RCCL group cannot come up) the probe must fail on every rank
    with a recorded error within its timeout, every rank must agree it failed, and the gloo group must keep working."""
    import torch.multiprocessing as mp
    mp.spawn(_probe_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = json.loads((tmp_path / ("probe.rank%d" % r)).read_text())
        assert got["res"]["ok"] is False and got["res"]["error"] and got["res"]["ranks_seen"] in (None, 1, 2)
        assert got["res"]["timeout_s"] == 20 and got["seconds"] < 60 and got["sum"] == 2.0
        assert got["skipped"]["ok"] is False and "skipped" in got["skipped"]["error"]
