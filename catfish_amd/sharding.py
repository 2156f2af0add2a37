"""Multi-GPU: reads shard across ranks, one process per GPU, no collective on the data path.

Every 35-sample window is independent (zero GRU state, window-local padding:
rnn_class.py:159,170-171) and every read is independent (catfish/catfish:55-56), so the
reference's sequential per-file loop partitions into disjoint shards.  Each rank runs its
shard on its own MI355X with its own copy of the 0.79 MB weights; the only communication is
the final HOST gather of the per-read results to rank 0 (pickled Python objects over a gloo
group) -- RCCL/xGMI is never on the critical path.

Entry points
    ``run_sharded_indexed``   the core: shard item INDICES by cost, run a callable on this rank's
                              indices, gather on rank 0 (a rank never touches another rank's items)
    ``run_sharded``           the same for a list of in-memory signals
    ``infer_reads_sharded``   product path: this rank's reads stream through one HipEngine's
                              ``ReadPipeline`` (int16 DAC up, spans down), results gathered on rank 0
    ``infer_files_sharded``   the reference's per-file loop (catfish/catfish:50-56): files shard by size,
                              every rank loads only its own files, a bounded batch at a time
Launch: ``python -m torch.distributed.run --nproc-per-node N -m catfish_amd.cli -i ... -s ...`` (or
``catfish_amd.cli --gpus N``, which starts that launcher as a child process before any GPU call).
"""
from __future__ import annotations

import os

import numpy as np

from .infer import WINDOW_SIZE, padding_size_for


def windows_of(length, window=WINDOW_SIZE):
    return (int(length) + padding_size_for(int(length), window)) // window


def shard_costs(costs, world_size):
    """Greedy longest-processing-time partition of item indices by cost.

    Returns ``world_size`` lists of indices (each ascending).  Equal costs degenerate to a
    round-robin deal.
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    cost = np.asarray(costs, dtype=np.int64)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += cost[i]
    return [sorted(s) for s in shards]


def shard_contiguous(costs, world_size):
    """Contiguous blocks of item indices with near-equal cost: rank r gets ``[cut[r], cut[r+1])`` where the cuts sit at
    the
    item boundaries nearest to the multiples of total / world_size of the running cost (a block is off by at most one
    item's cost).  Results gathered
    in rank order are then already in item order, which lets rank 0 concatenate tables without re-ordering them."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    cost = np.asarray(costs, dtype=np.int64)
    run = np.concatenate(([0], np.cumsum(cost)))
    want = run[-1] * np.arange(1, world_size) / float(world_size)
    hi = np.clip(np.searchsorted(run, want, side="left"), 1, len(cost)) if len(cost) else np.zeros(world_size - 1, np.int64)
    cuts = np.where(run[hi] - want < want - run[hi - 1], hi, hi - 1) if len(cost) else hi      # the nearer of the two cuts
    cuts = np.concatenate(([0], cuts, [len(cost)]))
    cuts = np.maximum.accumulate(cuts)
    return [list(range(int(cuts[r]), int(cuts[r + 1]))) for r in range(world_size)]


def shard_reads(lengths, world_size):
    """LPT partition of reads by window count (the unit of device work)."""
    return shard_costs([windows_of(n) for n in lengths], world_size)


def dist_env():
    """(rank, world_size, local_rank) from the launcher's environment (torch.distributed.run); 0, 1, 0 without one."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_host_group(timeout_s=None):
    """Initialise the process group of a sharded job when the launcher started more than one rank.

    gloo only: the data path has no collective, the group serves the final host gather (and barriers).
    ``timeout_s`` (default ``CATFISH_DIST_TIMEOUT_S`` or 1800): how long a rank waits in a collective for the others --
    it must cover the slowest rank's whole shard, because the fast ranks sit in the final gather meanwhile.
    Returns True when a group was created here (the caller then destroys it).
    """
    import datetime
    import torch.distributed as dist
    _rank, world, _local = dist_env()
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if timeout_s is None:
        timeout_s = float(os.environ.get("CATFISH_DIST_TIMEOUT_S", "1800"))
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(timeout_s)))
    return True


def agree_or_raise(error, stage, group=None, token=None):
    """Every rank reports how its ``stage`` went (``error`` = the exception it caught, or None) and all of them learn the
    outcome: a rank that failed re-raises its own exception, the others raise a RuntimeError naming it.  One small
    all-gather; call it after any per-rank step that can fail BEFORE the data path (creating directories, loading the
    network, opening the device), so that no rank walks into a later collective its peers will never reach.

    ``token`` (optional, any small picklable value) rides along: when no rank failed but the ranks' tokens differ, all of them
    raise -- for facts every rank derives on its own and all must agree on (the listing of the input directory)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if error is not None:
            raise error
        return
    said = [None] * dist.get_world_size()
    dist.all_gather_object(said, (None if error is None else "%s: %s" % (type(error).__name__, error), token), group=group)
    if error is not None:
        raise error
    bad = ["rank %d: %s" % (r, t) for r, (t, _tok) in enumerate(said) if t is not None]
    if bad:
        raise RuntimeError("%s failed on %s" % (stage, "; ".join(bad)))
    differing = [r for r, (_t, tok) in enumerate(said) if tok != said[0][1]]
    if differing:
        raise RuntimeError("%s: rank(s) %s disagree with rank 0 (%r against %r)" % (
            stage, ", ".join(str(r) for r in differing), said[differing[0]][1], said[0][1]))


class DirListing(object):
    """The input directory's entries in one agreed order, held by the library (``cf_listing_*``, include/catfish_hip.h): the
    reference's ``input_files = os.listdir(input_dir)`` (catfish/catfish:49-50) without a Python string per entry.  A rank asks it
    for the sizes of one block (to cut the list into equal work) and for the names of the block it classifies; the order is bytewise
    (= ``sorted()`` of the names whenever they are valid UTF-8), ``digest`` is what the ranks compare."""

    def __init__(self, directory):
        import ctypes as C
        from . import _native as N
        self._lib = N.lib()
        self._handle = C.c_void_p()
        n, dig = C.c_int64(0), (C.c_uint64 * 2)()
        N.check(self._lib.cf_listing_open(os.fsencode(directory), C.byref(self._handle), C.byref(n), dig))
        self.directory = directory
        self.n = int(n.value)
        self.digest = "%016x%016x" % (dig[0], dig[1])

    @classmethod
    def from_names_blob(cls, directory, blob, n_entries):
        """The listing another rank read: ``blob`` = the ordered names, NUL-terminated, back to back (``names_blob``)."""
        import ctypes as C
        from . import _native as N
        self = cls.__new__(cls)
        self._lib = N.lib()
        self._handle = C.c_void_p()
        dig = (C.c_uint64 * 2)()
        N.check(self._lib.cf_listing_from_names(os.fsencode(directory), blob, len(blob), int(n_entries), C.byref(self._handle), dig))
        self.directory, self.n, self.digest = directory, int(n_entries), "%016x%016x" % (dig[0], dig[1])
        return self

    def names_blob(self, lo=0, hi=None):
        """The names of entries [lo, hi) as bytes: NUL-terminated, back to back, in order (what travels between ranks)."""
        return self._names_raw(lo, self.n if hi is None else hi)[0]

    def _names_raw(self, lo, hi):
        import ctypes as C
        from . import _native as N
        need = C.c_int64(0)
        N.check(self._lib.cf_listing_names(self._handle, int(lo), int(hi), None, 0, None, C.byref(need)))
        if hi <= lo:
            return b"", np.zeros(1, dtype=np.int64)
        buf = C.create_string_buffer(max(1, int(need.value)))
        bounds = np.empty(hi - lo + 1, dtype=np.int64)
        N.check(self._lib.cf_listing_names(self._handle, int(lo), int(hi), buf, int(need.value), bounds.ctypes.data_as(C.c_void_p), None))
        return buf.raw[:int(need.value)], bounds

    def __len__(self):
        return self.n

    def sizes(self, lo, hi, n_threads=4):
        """int64 sizes on disk of entries [lo, hi)."""
        import ctypes as C
        from . import _native as N
        out = np.empty(max(0, hi - lo), dtype=np.int64)
        N.check(self._lib.cf_listing_sizes(self._handle, int(lo), int(hi), out.ctypes.data_as(C.c_void_p), int(n_threads)))
        return out

    def names(self, lo=0, hi=None):
        """The names of entries [lo, hi) as a list of str (this is where Python strings get built: ask for your block only)."""
        hi = self.n if hi is None else hi
        blob = self._names_raw(lo, hi)[0]
        return os.fsdecode(blob[:-1]).split("\x00") if hi > lo else []

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.cf_listing_close(self._handle)
            self._handle.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ListingPaths(object):
    """``directory/name`` of every entry of a ``DirListing`` as a sequence that builds strings only for the stretch in use:
    ``block(lo, hi)`` materialises the paths of [lo, hi) (a rank's files), indexing inside it is a list look-up."""

    def __init__(self, listing):
        self.listing = listing
        self._lo, self._paths = 0, []

    def __len__(self):
        return len(self.listing)

    def block(self, lo, hi):
        prefix = self.listing.directory.rstrip("/") + "/"
        self._lo, self._paths = int(lo), [prefix + n for n in self.listing.names(lo, hi)]
        return self

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if not self._lo <= i < self._lo + len(self._paths):
            self.block(i, min(len(self), i + 2048))
        return self._paths[i - self._lo]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def stat_sizes(directory, names, n_threads=4):
    """Sizes on disk of ``names`` (entries of ``directory``) as int64 -- ``cf_stat_files``: fstatat from the library's host thread
    pool (a Python loop of os.stat costs 1.5 us per file, 19 ms for a rank's 12 500 reads; this about half).  A name that cannot
    be stat-ed raises ValueError naming it."""
    import ctypes as C
    from . import _native as N
    sizes = np.empty(len(names), dtype=np.int64)
    if not names:
        return sizes
    enc = [os.fsencode(n) for n in names]
    blob = b"\x00".join(enc) + b"\x00"
    bounds = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(np.fromiter((len(e) + 1 for e in enc), dtype=np.int64, count=len(enc)), out=bounds[1:])
    N.check(N.lib().cf_stat_files(os.fsencode(directory), blob, bounds.ctypes.data_as(C.c_void_p), len(enc),
                                  sizes.ctypes.data_as(C.c_void_p), int(n_threads)))
    return sizes


def _host_threads():
    """Threads for the library's host pools (stat, file reads): a quarter of the CPUs this rank owns (``placement.bind``), 2 .. 8."""
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    return max(2, min(8, n_cpu // 4))


def shared_listing(input_dir, rank=0, world_size=1, group=None, error=None):
    """The listing step of the sharded per-file loop (catfish/catfish:49-50 lists the directory once, in one process) ->
    ``(listing, sizes)``: a ``DirListing`` over ALL entries in one order and their int64 sizes on disk, the same on every rank.

    RANK 0 READS THE DIRECTORY, once, and broadcasts the ordered names (1.6 MB for 100 000 reads); every rank builds its listing
    from them and stats only entries ``[rank n / N, (rank + 1) n / N)`` (fstatat from the library's threads); one all-gather
    completes the sizes.  Why one reader: concurrent readdirs of ONE directory serialise on some file systems -- on the MI355X boxes'
    overlayfs 8 ranks x 100 000 entries took 102 ms EACH against 13 ms for a single reader (tools/exp_listing.py,
    profiles/r05_listing_concurrency.log) -- so N readers cost N times the one reader everybody would otherwise wait for.

    ``error``: the exception of a step before (rank 0 could not create the output directories): it travels in the listing's place.
    Failure anywhere -- rank 0 cannot read the directory, a rank cannot stat an entry of its block (gone since the listing, or not
    visible from that rank: a file still being copied in, a stale network file system) -- raises on EVERY rank: the failing one
    its own exception, the others a RuntimeError naming rank and cause; nobody walks on with a list the others do not share."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    listing = None
    if rank == 0 and error is None:
        try:
            listing = DirListing(input_dir)
        except Exception as exc:              # noqa: BLE001 -- told to the other ranks below
            error = exc
    if not distributed:
        if error is not None:
            raise error
        return listing, listing.sizes(0, len(listing), n_threads=_host_threads())
    box = [None]
    if rank == 0:
        box[0] = ("error", "%s: %s" % (type(error).__name__, error)) if error is not None else (len(listing), listing.names_blob())
    dist.broadcast_object_list(box, src=0, group=group)
    sizes = lo = None
    if box[0][0] != "error" and error is None:
        try:
            if rank != 0:
                listing = DirListing.from_names_blob(input_dir, box[0][1], box[0][0])
            lo, hi = rank * len(listing) // world_size, (rank + 1) * len(listing) // world_size
            sizes = listing.sizes(lo, hi, n_threads=_host_threads())
        except Exception as exc:              # noqa: BLE001
            error = exc
    said = [None] * dist.get_world_size()
    dist.all_gather_object(said, (None if error is None else "%s: %s" % (type(error).__name__, error), lo, sizes), group=group)
    if error is not None:
        raise error
    stage = "listing the input directory"
    if box[0][0] == "error":
        raise RuntimeError("%s failed on rank 0: %s" % (stage, box[0][1]))
    bad = ["rank %d: %s" % (r, t[0]) for r, t in enumerate(said) if t[0] is not None]
    if bad:
        raise RuntimeError("%s failed on %s" % (stage, "; ".join(bad)))
    all_sizes = np.concatenate([t[2] for t in said])
    if [t[1] for t in said] != [r * len(listing) // len(said) for r in range(len(said))] or len(all_sizes) != len(listing):
        raise RuntimeError("%s: the ranks' blocks do not tile the listing" % stage)
    return listing, all_sizes


def host_gather_group():
    """A gloo group for the final host gather (object pickles travel over TCP/shared memory, never
    through RCCL); falls back to the default group when that already is gloo."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend() == "gloo":
        return None
    return dist.new_group(backend="gloo")


class SpanTable(object):
    """Results of many reads as four arrays instead of nested Python lists: what a rank hands to the host gather (numpy
    arrays pickle at memcpy speed; a list of lists of ints does not).

    read_of [n_spans]  position of the span's read in this table (ascending)
    start, end [n_spans]  the reference's span coordinates (start - 11, end + 16; infer.py:141-162)
    lengths [n_reads]  real length of every read
    """

    def __init__(self, read_of, start, end, lengths):
        self.read_of = np.asarray(read_of, dtype=np.int64)
        self.start = np.asarray(start, dtype=np.int64)
        self.end = np.asarray(end, dtype=np.int64)
        self.lengths = np.asarray(lengths, dtype=np.int64)

    def __len__(self):
        return int(self.lengths.shape[0])

    @classmethod
    def from_lists(cls, results):
        read_of = [i for i, (spans, _n) in enumerate(results) for _ in spans]
        flat = [sp for spans, _n in results for sp in spans]
        arr = np.asarray(flat, dtype=np.int64).reshape(-1, 2)
        return cls(read_of, arr[:, 0], arr[:, 1], [n for _spans, n in results])

    @classmethod
    def concat(cls, tables):
        if not tables:
            return cls([], [], [], [])
        offs = np.cumsum([0] + [len(t) for t in tables[:-1]])
        return cls(np.concatenate([t.read_of + o for t, o in zip(tables, offs)]), np.concatenate([t.start for t in tables]),
                   np.concatenate([t.end for t in tables]), np.concatenate([t.lengths for t in tables]))

    def read(self, r):
        """-> (spans, length) of read ``r`` alone, the reference's return value for it (infer.py:12-51)."""
        lo, hi = np.searchsorted(self.read_of, [r, r + 1])
        return np.stack([self.start[lo:hi], self.end[lo:hi]], axis=1).tolist(), int(self.lengths[r])

    def expand(self):
        """-> [(spans, length)] per read, the reference's return values (one ``tolist`` + one slice per read)."""
        from .batching import quiet_gc
        n = len(self)
        with quiet_gc():          # ~20 small lists per read: with the cyclic collector on, every 700 of them walk the whole process
            pairs = np.stack([self.start, self.end], axis=1).tolist()
            bounds = np.concatenate(([0], np.cumsum(np.bincount(self.read_of, minlength=n)[:n]))).tolist()
            lens = self.lengths.tolist()
            return [(pairs[bounds[r]:bounds[r + 1]], lens[r]) for r in range(n)]


def run_sharded_indexed(costs, work_fn, rank=None, world_size=None, gather_group=None, partition="lpt", assemble=None):
    """Shard ``len(costs)`` items by cost, run ``work_fn(list of my indices) -> list of results`` on this
    rank's shard, gather on rank 0.

    Returns the full result list in item order on rank 0 and None on the other ranks.  Without an
    initialised process group it simply runs everything locally.

    ``partition``: "lpt" (greedy longest-processing-time deal, ``shard_costs``) or "contiguous" (``shard_contiguous``).
    ``assemble``: when given, ``work_fn`` may return any picklable object with a length and rank 0 returns
    ``assemble([(indices, result) of rank 0, of rank 1, ...])`` instead of a per-item list -- the way tables stay tables.
    """
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if distributed else 0
    if world_size is None:
        world_size = dist.get_world_size() if distributed else 1
    n_items = len(costs)
    if partition not in ("lpt", "contiguous"):
        raise ValueError("partition must be 'lpt' or 'contiguous'")
    shards = shard_costs(costs, world_size) if partition == "lpt" else shard_contiguous(costs, world_size)
    mine = shards[rank]
    # A rank whose shard fails (unreadable file, device error) must still reach the gather, or the others would wait
    # for it until the process group times out: its error text travels in place of its results and every rank raises.
    failure = None
    local = []
    try:
        local = work_fn(mine) if (mine or assemble is not None) else []
        if assemble is None and not isinstance(local, SpanTable):
            local = list(local)
        if len(local) != len(mine):
            raise RuntimeError("work_fn returned %d results for %d items" % (len(local), len(mine)))
    except Exception as exc:          # noqa: BLE001 -- re-raised below, after the collective
        if world_size == 1 or not distributed:
            raise
        failure = exc

    def place(out, idx, res):
        for i, r in zip(idx, res.expand() if isinstance(res, SpanTable) else res):
            out[i] = r

    if world_size == 1:
        if assemble is not None:
            return assemble([(mine, local)])
        out = [None] * n_items
        place(out, mine, local)
        return out
    if not distributed:
        raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group")
    gathered = [None] * world_size if rank == 0 else None
    payload = (mine, local) if failure is None else ("error", "%s: %s" % (type(failure).__name__, failure))
    dist.gather_object(payload, gathered, dst=0, group=gather_group)            # a SpanTable travels as four arrays
    verdict = [None]
    if rank == 0:
        bad = ["rank %d: %s" % (r, g[1]) for r, g in enumerate(gathered) if isinstance(g[0], str) and g[0] == "error"]
        verdict[0] = "; ".join(bad) if bad else ""
    dist.broadcast_object_list(verdict, src=0, group=gather_group)              # every rank learns whether the job failed
    if failure is not None:
        raise failure
    if verdict[0]:
        raise RuntimeError("sharded run failed on " + verdict[0])
    if rank != 0:
        return None
    if assemble is not None:
        return assemble(gathered)
    out = [None] * n_items
    for idx, res in gathered:
        place(out, idx, res)
    return out


def run_sharded(signals, infer_fn, rank=None, world_size=None, gather_group=None):
    """Run ``infer_fn(list of signals) -> list of results`` on this rank's shard and gather on rank 0.

    All ranks pass the same ``signals`` list (or at least the same lengths: a rank only touches
    its own shard's entries).
    """
    costs = [windows_of(len(s)) if s is not None else 0 for s in signals]
    return run_sharded_indexed(costs, lambda mine: infer_fn([signals[i] for i in mine]),
                               rank=rank, world_size=world_size, gather_group=gather_group)


# --------------------------------------------------------------------------- product path
RAMP = (0.125, 0.375)      # the first batches of a file-driven shard, as fractions of the full batch (see _batches_by_samples)
if os.environ.get("CATFISH_DEBUG_KNOBS", "0") not in ("", "0") and os.environ.get("CATFISH_RAMP") == "0":
    RAMP = ()                # A/B knob for tools/


def _batches_by_samples(indices, lengths, max_samples, ramp=()):
    """Consecutive groups of ``indices`` holding at most ``max_samples`` samples each (at least one read).

    ``ramp``: fractions of ``max_samples`` that cap the FIRST batches instead -- a shard that starts from files has nothing on
    the GPU until its first batch is read, staged and uploaded (1110 reads of 4096 samples: ~4 ms); with a first batch of an
    eighth the device starts after ~0.5 ms and the loader thread has the next, larger batches ready before it runs dry.  Windows
    are independent, so how reads are grouped never changes a result."""
    cur, tot, k = [], 0, 0
    for i in indices:
        n = int(lengths[i])
        cap = max_samples * ramp[k] if k < len(ramp) else max_samples
        if cur and tot + n > cap:
            yield cur
            cur, tot, k = [], 0, k + 1
        cur.append(i)
        tot += n
    if cur:
        yield cur


class EngineBatchRunner(object):
    """This rank's device work: batches of raw reads through ONE engine.

    int16 DAC reads take the streaming ``ReadPipeline`` (2 B/sample up, device median/MAD normalisation,
    double-buffered); anything else (float signals, e.g. an already normalised trace) is normalised on the
    host like ``infer.process_signal`` and goes through ``batching.infer_reads``.
    """

    def __init__(self, model, max_samples_per_batch, threshold=0.5, min_run=15):
        from .pipeline import ReadPipeline
        self.engine = model.engine if hasattr(model, "engine") else model
        if self.engine is None:
            raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
        self.max_samples = int(max_samples_per_batch)
        self.threshold, self.min_run = threshold, min_run
        self.pipe = ReadPipeline(self.engine, self.max_samples, threshold=threshold, min_run=min_run)

    def run(self, batches, compact=False):
        """``batches``: iterable of lists of raw reads -> yields per batch, in order, ``[(spans, length)]`` or (``compact``)
        a ``SpanTable`` of the batch."""
        self.compact = bool(compact)
        return self._drive(self._submit_reads(reads) for reads in batches)

    def run_files(self, path_batches, compact=False):
        """The same from batches of FILE NAMES: int16 ``.npy`` reads go from disk into the pipeline's pinned staging buffer
        through the library's host thread pool (``ReadPipeline.submit_files``); a batch that needs the general loader (other
        formats, or more samples than the size on disk suggested) is loaded with ``infer.load_dac`` and re-cut by its true
        lengths.  Results come out in the order of the names, one item per submitted batch."""
        self.compact = bool(compact)
        from .infer import load_dac

        def items():
            for paths in path_batches:
                ticket = self.pipe.submit_files(paths)
                if ticket is not None:
                    yield (ticket, None)
                    continue
                reads = [load_dac(p) for p in paths]
                for idx in _batches_by_samples(list(range(len(reads))), [_padded(len(r)) for r in reads], self.max_samples):
                    yield self._submit_reads([reads[i] for i in idx])
        return self._drive(items())

    def run_listing(self, paths, ranges, compact=False):
        """``run_files`` for index ranges of a ``ListingPaths``: batch [lo, hi) goes from the listing's directory into the pinned
        staging buffer without a path string per file (``ReadPipeline.submit_listing``); a batch that needs the general loader is
        handled like in ``run_files``, from the paths of just that batch."""
        if getattr(self, "pipe", None) is None or type(self).run_files is not EngineBatchRunner.run_files:
            # a runner that brings its own file handling (tests stand the oracle in for the engine): hand it the paths
            return self.run_files(([paths[i] for i in range(lo, hi)] for lo, hi in ranges), compact)
        self.compact = bool(compact)
        from .infer import load_dac

        def items():
            # the files of batch k + 1 are read (helper thread, ``preload_listing``) while the main thread waits for the GPU and
            # assembles batch k - 1: a preload starts right after the launch before it and is launched first thing afterwards
            pre = None
            try:
                todo = iter(ranges)
                nxt = next(todo, None)
                if nxt is not None:
                    pre = self.pipe.preload_listing(paths.listing, *nxt)
                while nxt is not None:
                    lo, hi = nxt
                    mine, pre = pre, None
                    ticket = self.pipe.launch_preloaded(mine)
                    nxt = next(todo, None)
                    if ticket is not None:
                        if nxt is not None:
                            pre = self.pipe.preload_listing(paths.listing, *nxt)
                        yield (ticket, None)
                        continue
                    reads = [load_dac(paths[i]) for i in range(lo, hi)]          # general loader, nothing preloading meanwhile
                    for idx in _batches_by_samples(list(range(len(reads))), [_padded(len(r)) for r in reads], self.max_samples):
                        yield self._submit_reads([reads[i] for i in idx])
                    if nxt is not None:
                        pre = self.pipe.preload_listing(paths.listing, *nxt)
            finally:
                self.pipe.drop_preloaded(pre)
        return self._drive(items())

    def _submit_reads(self, reads):
        from . import batching
        from .infer import is_dac, normalize_raw_signal
        if all(is_dac(r) for r in reads):
            return (self.pipe.submit([np.ascontiguousarray(r, dtype=np.int16) for r in reads]), None)
        normed = [normalize_raw_signal(np.asarray(r), "median") for r in reads]
        max_windows = max(1, self.max_samples // WINDOW_SIZE)
        return (None, batching.infer_reads(self.engine, normed, max_windows=max_windows,
                                           threshold=self.threshold, min_run=self.min_run))

    def _drive(self, items):
        """Up to the pipeline's depth of batches in flight: ``items`` submits lazily (each element is (ticket, host-path results
        or None)), results are finished in order."""
        from collections import deque
        pending = deque()
        for item in items:
            pending.append(item)
            if len(pending) == self.pipe.depth:
                yield self._finish(pending.popleft())
        while pending:
            yield self._finish(pending.popleft())

    def _finish(self, item):
        ticket, host_res = item
        if ticket is None:
            return SpanTable.from_lists(host_res) if self.compact else host_res
        if not self.compact:
            return self.pipe.collect(ticket)
        read_of, start, end, lengths = self.pipe.collect(ticket, as_lists=False)
        return SpanTable(read_of, start, end, lengths)


def _prefetched(gen, depth=3):
    """Run a generator of read batches in a loader thread, ``depth`` batches ahead: file reads (which release the GIL)
    and array assembly overlap the main thread's waits on the GPU.  Exceptions of the loader are re-raised here."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()

    def loader():
        try:
            for item in gen:
                q.put((item, None))
            q.put((done, None))
        except BaseException as exc:           # noqa: BLE001 -- handed to the consumer
            q.put((done, exc))

    t = threading.Thread(target=loader, name="catfish-loader", daemon=True)
    t.start()
    while True:
        item, exc = q.get()
        if item is done:
            t.join()
            if exc is not None:
                raise exc
            return
        yield item


def _spans_of_shard(model, reads, mine, lengths, load_fn, max_samples_per_batch, batch_runner, compact, size_hints=None):
    """This rank's device work: the reads ``mine`` (indices into ``reads``) through the batch runner, batches cut at
    ``max_samples_per_batch`` samples, files loaded a bounded number of batches ahead.  ``compact``: a ``SpanTable`` (arrays
    end to end) instead of ``[(spans, length)]`` lists.  ``size_hints`` (estimated samples per item, e.g. from the size on
    disk): ``reads`` are file names for ``infer.load_dac`` and the runner may read them itself (``run_files``: the library's
    native loader straight into pinned memory instead of one Python call per file)."""
    runner = batch_runner if batch_runner is not None else EngineBatchRunner(model, max_samples_per_batch)
    if size_hints is not None and isinstance(reads, ListingPaths) and hasattr(runner, "run_listing"):
        # contiguous indices of a native listing: the batches travel as (lo, hi), the names stay in the library
        ranges = ((idx[0], idx[-1] + 1) for idx in _batches_by_samples(mine, size_hints, max_samples_per_batch, ramp=RAMP))
        if compact:
            return SpanTable.concat(list(runner.run_listing(reads, ranges, compact=True)))
        out = []
        for res in runner.run_listing(reads, ranges):
            out.extend(res)
        return out
    if size_hints is not None and hasattr(runner, "run_files"):
        path_batches = ([reads[i] for i in idx] for idx in _batches_by_samples(mine, size_hints, max_samples_per_batch, ramp=RAMP))
        if compact:
            return SpanTable.concat(list(runner.run_files(path_batches, compact=True)))
        out = []
        for res in runner.run_files(path_batches):
            out.extend(res)
        return out

    def batches():
        if lengths is not None:
            for idx in _batches_by_samples(mine, lengths, max_samples_per_batch):
                yield [reads[i] if load_fn is None else load_fn(reads[i]) for i in idx]
        else:                       # lengths unknown until loaded: cut a batch when the next read would not fit
            cur, tot = [], 0
            for i in mine:
                r = load_fn(reads[i])
                if cur and tot + len(r) > max_samples_per_batch:
                    yield cur
                    cur, tot = [], 0
                cur.append(r)
                tot += len(r)
            if cur:
                yield cur
    source = _prefetched(batches(), depth=3) if load_fn is not None else batches()
    if compact:
        if isinstance(runner, EngineBatchRunner):
            return SpanTable.concat(list(runner.run(source, compact=True)))
        return SpanTable.concat([SpanTable.from_lists(res) for res in runner.run(source)])
    # the per-read lists of batch k are built while the GPU runs batch k + 1 (0.3 ms per batch with the cyclic collector
    # off, see batching.quiet_gc)
    out = []
    for res in runner.run(source):
        out.extend(res)
    return out


def infer_reads_sharded(model, reads, lengths=None, load_fn=None, max_samples_per_batch=None, batch_runner=None,
                        rank=None, world_size=None, gather_group=None, costs=None, size_hints=None, as_table=False):
    """Homopolymer spans of many reads, sharded over the ranks of the job; rank 0 gets ``[(spans, length)]``
    in input order, the other ranks get None.

    ``reads``      list of raw reads (int16 DAC arrays) -- or of anything ``load_fn(item)`` turns into one
                   (e.g. file paths: a rank only loads its own shard, one bounded batch at a time)
    ``lengths``    per-read sample counts when known up front (default ``len(read)``); with ``load_fn`` and
                   no lengths the batches are cut after loading
    ``costs``      sharding weights (default: window counts from ``lengths``)
    ``batch_runner`` object with ``run(iterable of read batches) -> iterable of per-batch result lists``
                   (default: ``EngineBatchRunner(model)`` = the HIP engine of this rank)

    This is the reference's RESULT TYPE (Python lists per read): ~20 list objects per read built on rank 0, which at 8 ranks
    costs about as much as the classification itself.  ``as_table=True`` keeps the results flat instead: reads go to ranks in
    contiguous blocks and rank 0 returns ONE ``SpanTable`` over all reads in input order (``table.read(i)`` /
    ``table.expand()`` give the lists when wanted).  A caller that goes on to merge the spans into chunks wants
    ``chunk_files_local`` / ``cli.run_pipeline``, where nothing leaves the ranks at all.
    """
    n = len(reads)
    if lengths is None and load_fn is None:
        lengths = [len(r) for r in reads]
    if costs is None:
        if lengths is None:
            raise ValueError("infer_reads_sharded: pass lengths or costs together with load_fn")
        costs = [windows_of(x) for x in lengths]
    if max_samples_per_batch is None:
        max_samples_per_batch = 32768 * WINDOW_SIZE
    if n == 0:
        empty = SpanTable([], [], [], []) if as_table else []
        return empty if (dist_env()[0] if rank is None else rank) == 0 else None

    def work(mine):
        import torch.distributed as dist
        n_ranks = world_size if world_size is not None else (
            dist.get_world_size() if dist.is_available() and dist.is_initialized() else dist_env()[1])
        # several ranks: arrays all the way to the gather (they pickle at memcpy speed), lists are built on rank 0
        compact = as_table or (n_ranks > 1 and (batch_runner is None or isinstance(batch_runner, EngineBatchRunner)))
        return _spans_of_shard(model, reads, mine, lengths, load_fn, max_samples_per_batch, batch_runner, compact,
                               size_hints=size_hints)

    from .batching import quiet_gc
    with quiet_gc():
        if as_table:
            return run_sharded_indexed(costs, work, rank=rank, world_size=world_size, gather_group=gather_group,
                                       partition="contiguous", assemble=lambda got: SpanTable.concat([t for _idx, t in got]))
        return run_sharded_indexed(costs, work, rank=rank, world_size=world_size, gather_group=gather_group)


def _file_costs(paths):
    """Size on disk of every file (1 for a missing one: the loader reports it when its turn comes) -- one stat each."""
    out = []
    for p in paths:
        try:
            out.append(max(1, os.stat(p).st_size))
        except OSError:
            out.append(1)
    return out


def _padded(n, window=WINDOW_SIZE):
    """Samples a read of ``n`` samples occupies in a packed launch (infer.py:32-36: whole windows, a multiple gets an extra one)."""
    return (int(n) // window + 1) * window


def _sample_hints(file_sizes):
    """What a file will occupy in a packed launch, estimated from its size on disk: its samples (int16 behind numpy's usual 128-byte
    header) rounded up to whole windows the way ``infer.py:32-36`` pads them (a multiple of 35 gets a full extra window).  Only used
    to cut batches before anything is read -- a batch that turns out too big is re-cut by its true lengths.  Counting PADDED samples
    matters: a batch is capped at the engine's windows per pass, and 1120 reads of 4096 samples are 132 160 windows, not 131 072 --
    every full batch used to spill 1088 windows into a second, nearly empty forward pass (0.42 of 2.9 ms per batch in bf16; kernel trace
    of round 5)."""
    n = np.maximum(1, (np.asarray(file_sizes, dtype=np.int64) - 128) // 2)
    return (n // WINDOW_SIZE + 1) * WINDOW_SIZE


def infer_files_sharded(model, paths, max_samples_per_batch=None, batch_runner=None, rank=None, world_size=None,
                        gather_group=None):
    """The reference's per-file loop (catfish/catfish:50-56), sharded: files are dealt to ranks by size on disk
    (a proxy of the sample count that needs no read), every rank loads only its own files through
    ``infer.load_dac`` -- one batch at a time, so host memory does not grow with the directory -- and rank 0
    receives ``[(spans, read length)]`` in the order of ``paths``."""
    from .infer import load_dac
    costs = _file_costs(paths)
    return infer_reads_sharded(model, list(paths), lengths=None, load_fn=load_dac, costs=costs,
                               max_samples_per_batch=max_samples_per_batch, batch_runner=batch_runner,
                               rank=rank, world_size=world_size, gather_group=gather_group, size_hints=_sample_hints(costs))


def chunk_files_local(model, paths, chunk_size=1000, max_samples_per_batch=None, batch_runner=None, rank=None,
                      world_size=None, timings=None, file_sizes=None):
    """This rank's share of the reference's per-file loop body (catfish/catfish:55-82), NO collective: the files are cut into
    ``world_size`` contiguous blocks of near-equal size on disk (``shard_contiguous``), this rank classifies block ``rank`` and
    merges / centres / complements its spans (``chunks.ChunkTable``: one native call over the rank's span table).
    -> (indices of my files in ``paths`` -- a contiguous range --, their ChunkTable).  ``timings``: optional dict that receives
    ``infer_s`` / ``chunks_s``.  ``file_sizes``: sizes on disk when the caller has them already (a directory scan yields them for
    free; otherwise one ``stat`` per file here)."""
    import time
    from .chunks import ChunkTable
    from .infer import load_dac
    lazy = isinstance(paths, ListingPaths)              # a big directory: path strings only for this rank's block
    paths = paths if lazy else list(paths)
    if max_samples_per_batch is None:
        max_samples_per_batch = 32768 * WINDOW_SIZE
    env_rank, env_world, _local = dist_env()
    rank = env_rank if rank is None else int(rank)
    world_size = env_world if world_size is None else int(world_size)
    costs = np.maximum(1, np.asarray(file_sizes, dtype=np.int64)) if file_sizes is not None else _file_costs(paths)
    if len(costs) != len(paths):
        raise ValueError("file_sizes must hold one size per path")
    mine = shard_contiguous(costs, world_size)[rank]
    from .batching import quiet_gc
    with quiet_gc():
        t0 = time.perf_counter()
        table = _spans_of_shard(model, paths, mine, None, load_dac, max_samples_per_batch, batch_runner, True,
                                size_hints=_sample_hints(costs)) if mine else SpanTable([], [], [], [])
        t1 = time.perf_counter()
        out = ChunkTable.from_span_table(table, chunk_size)
    if timings is not None:
        timings["infer_s"], timings["chunks_s"] = t1 - t0, time.perf_counter() - t1
    if len(out) != len(mine):
        raise RuntimeError("chunk_files_local: %d results for %d files" % (len(out), len(mine)))
    return mine, out


def chunk_files_sharded(model, paths, chunk_size=1000, max_samples_per_batch=None, batch_runner=None, rank=None,
                        world_size=None, gather_group=None, timings=None, file_sizes=None):
    """``chunk_files_local`` on every rank + a host gather: what travels to rank 0 is six small arrays per rank and what rank 0
    does is concatenate them (the blocks are contiguous, so the gathered tables are already in the order of ``paths``).
    -> ``ChunkTable`` over all files on rank 0, None elsewhere.  ``timings`` also receives ``assemble_s`` on rank 0.  A caller
    that only needs the documents written (the CLI) skips this gather altogether: ``cli.run_pipeline``."""
    import time
    from .chunks import ChunkTable
    paths = list(paths)
    timings = {} if timings is None else timings
    costs = [max(1, int(c)) for c in file_sizes] if file_sizes is not None else _file_costs(paths)
    if len(costs) != len(paths):
        raise ValueError("file_sizes must hold one size per path")

    def work(mine):
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized()
        r = rank if rank is not None else (dist.get_rank() if distributed else 0)
        w = world_size if world_size is not None else (dist.get_world_size() if distributed else 1)
        got, out = chunk_files_local(model, paths, chunk_size, max_samples_per_batch, batch_runner, rank=r, world_size=w,
                                     timings=timings, file_sizes=costs)
        if got != list(mine):
            raise RuntimeError("chunk_files_sharded: shard mismatch")
        return out

    def assemble(gathered):
        t0 = time.perf_counter()
        seen = [i for idx, _t in gathered for i in idx]
        if seen != list(range(len(paths))):
            raise RuntimeError("chunk_files_sharded: shards do not tile the file list")
        out = ChunkTable.concat([t for _idx, t in gathered])
        timings["assemble_s"] = time.perf_counter() - t0
        return out

    return run_sharded_indexed(costs, work, rank=rank, world_size=world_size, gather_group=gather_group,
                               partition="contiguous", assemble=assemble)
