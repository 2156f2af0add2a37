"""The pipeline tail over flat tables: homopolymer spans of many reads -> merged chunks + their complement -> JSON.

The reference does this read by read inside its per-file loop (catfish/catfish:57-82, ``center_hp`` :121-135), i.e. it
is part of what shards over the GPUs' host processes: every rank turns the spans of ITS reads into a ``ChunkTable`` with
one native call (``cf_chunks_from_spans``, include/catfish_hip.h -- plain host code in the C-ABI library) and hands rank 0
six small arrays.  Nothing here builds per-span Python objects unless a caller asks for the reference's dicts
(``ChunkTable.to_dicts``).  ``cli.merge_positions`` / ``center_hp`` / ``nonhp_complement`` are the same rules in Python,
pinned by goldens made from the reference's own functions; ``tests/test_host_logic.py`` checks the native path against
them.
"""
from __future__ import annotations

import ctypes as C
import json
import re

import numpy as np

from . import _native as N

_PLAIN_KEYS = re.compile(r'[^\x20-\x7e]|["\\]')


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class ChunkTable(object):
    """Merged homopolymer chunks and non-homopolymer stretches of ``n_reads`` reads, CSR style.

    lengths       int64 [n_reads]      samples per read (second return value of infer_class_from_signal)
    hp_bounds     int64 [n_reads + 1]  read r owns rows [hp_bounds[r], hp_bounds[r+1]) of hp_start / hp_end
    nonhp_bounds  likewise for nonhp_start / nonhp_end; a read WITHOUT homopolymer rows holds one row (0, length)
    """

    def __init__(self, lengths, hp_bounds, hp_start, hp_end, nonhp_bounds, nonhp_start, nonhp_end):
        self.lengths = _i64(lengths)
        self.hp_bounds, self.hp_start, self.hp_end = _i64(hp_bounds), _i64(hp_start), _i64(hp_end)
        self.nonhp_bounds, self.nonhp_start, self.nonhp_end = _i64(nonhp_bounds), _i64(nonhp_start), _i64(nonhp_end)

    def __len__(self):
        return int(self.lengths.shape[0])

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_spans(cls, span_bounds, span_start, span_end, lengths, chunk_size=1000):
        """Spans as a CSR table (``span_bounds`` [n_reads + 1]) -> chunks, catfish/catfish:57-82 for every read."""
        span_bounds, span_start, span_end, lengths = (_i64(a) for a in (span_bounds, span_start, span_end, lengths))
        n_reads = int(lengths.shape[0])
        if span_bounds.shape[0] != n_reads + 1:
            raise ValueError("span_bounds must have n_reads + 1 entries")
        n_spans = int(span_bounds[-1] - span_bounds[0]) if n_reads else 0
        if span_start.shape[0] < n_spans or span_end.shape[0] < n_spans:
            raise ValueError("span table shorter than span_bounds says")
        hp_cap, non_cap = n_spans + n_reads, n_spans + 2 * n_reads
        hp_b, non_b = np.zeros(n_reads + 1, np.int64), np.zeros(n_reads + 1, np.int64)
        hp_s, hp_e = np.empty(hp_cap, np.int64), np.empty(hp_cap, np.int64)
        non_s, non_e = np.empty(non_cap, np.int64), np.empty(non_cap, np.int64)
        N.check(N.lib().cf_chunks_from_spans(_ptr(span_bounds), _ptr(span_start), _ptr(span_end), _ptr(lengths), n_reads,
                                             int(chunk_size), _ptr(hp_b), _ptr(hp_s), _ptr(hp_e), hp_cap,
                                             _ptr(non_b), _ptr(non_s), _ptr(non_e), non_cap))
        return cls(lengths, hp_b, hp_s[:hp_b[-1]], hp_e[:hp_b[-1]], non_b, non_s[:non_b[-1]], non_e[:non_b[-1]])

    @classmethod
    def from_span_table(cls, table, chunk_size=1000):
        """From a ``sharding.SpanTable`` (read_of ascending)."""
        n = len(table)
        bounds = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(table.read_of, minlength=n)[:n], out=bounds[1:])
        return cls.from_spans(bounds, table.start, table.end, table.lengths, chunk_size)

    @classmethod
    def concat(cls, tables):
        tables = list(tables)
        if not tables:
            return cls([], [0], [], [], [0], [], [])

        def join_bounds(parts):
            offs = np.cumsum([0] + [int(p[-1]) for p in parts[:-1]])
            return np.concatenate([parts[0][:1] * 0] + [p[1:] + o for p, o in zip(parts, offs)])

        return cls(np.concatenate([t.lengths for t in tables]),
                   join_bounds([t.hp_bounds for t in tables]), np.concatenate([t.hp_start for t in tables]),
                   np.concatenate([t.hp_end for t in tables]),
                   join_bounds([t.nonhp_bounds for t in tables]), np.concatenate([t.nonhp_start for t in tables]),
                   np.concatenate([t.nonhp_end for t in tables]))

    def take(self, order):
        """The table with its reads re-ordered / selected: row r of the result is read ``order[r]`` of this one."""
        order = _i64(order)

        def gather(bounds, *cols):
            counts = (bounds[1:] - bounds[:-1])[order]
            nb = np.zeros(len(order) + 1, np.int64)
            np.cumsum(counts, out=nb[1:])
            rows = np.repeat(bounds[:-1][order] - nb[:-1], counts) + np.arange(int(nb[-1]), dtype=np.int64)
            return (nb,) + tuple(c[rows] for c in cols)

        hp = gather(self.hp_bounds, self.hp_start, self.hp_end)
        non = gather(self.nonhp_bounds, self.nonhp_start, self.nonhp_end)
        return ChunkTable(self.lengths[order], *(hp + non))

    # ------------------------------------------------------------------ views
    @property
    def has_hp(self):
        return self.hp_bounds[1:] > self.hp_bounds[:-1]

    def json_members(self, names):
        """-> (hp_text, nonhp_text): the members of the two JSON objects (``{`` + text + ``}`` is the document), byte for byte
        what ``json.dump`` writes for the reference's ``hp_dict`` / ``nonhp_dict``.  ``names``: one per read."""
        names = list(names)
        if len(names) != len(self):
            raise ValueError("one name per read")
        if _PLAIN_KEYS.search("".join(names)) is None:         # printable ASCII without quote / backslash: no escaping needed
            blob = ('"' + '""'.join(names) + '"').encode("ascii") if names else b""
            key_len = np.fromiter((len(n) for n in names), dtype=np.int64, count=len(names)) + 2
        else:
            enc = [json.dumps(n).encode("ascii") for n in names]
            blob = b"".join(enc)
            key_len = np.array([len(e) for e in enc], dtype=np.int64)
        key_bounds = np.zeros(len(names) + 1, np.int64)
        np.cumsum(key_len, out=key_bounds[1:])
        keys = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, np.uint8)
        whole = np.ascontiguousarray(~self.has_hp, dtype=np.uint8)
        out = []
        for bounds, start, end, flag in ((self.hp_bounds, self.hp_start, self.hp_end, None),
                                         (self.nonhp_bounds, self.nonhp_start, self.nonhp_end, whole)):
            cap = int(key_bounds[-1]) + 48 * int(bounds[-1]) + 40 * len(names) + 64
            buf = np.empty(cap, np.uint8)
            n = N.lib().cf_chunks_json(_ptr(keys), _ptr(key_bounds), len(names), _ptr(bounds), _ptr(start), _ptr(end),
                                       _ptr(flag) if flag is not None else None, _ptr(buf), cap)
            if n < 0:
                N.check(int(n))
            out.append(buf[:n].tobytes())
        return tuple(out)

    def to_dicts(self, names):
        """The reference's two dicts (catfish/catfish:66-82): ``hp_dict[name] = [[start, end], ...]`` for reads with
        homopolymers, ``nonhp_dict[name]`` for every read (``[([(0, len), len])]`` when it has none).  Builds Python lists: for
        callers that want the reference's objects, not for the data path."""
        hp_pairs = np.stack([self.hp_start, self.hp_end], axis=1).tolist()
        non_pairs = np.stack([self.nonhp_start, self.nonhp_end], axis=1).tolist()
        hb, nb, lens = self.hp_bounds.tolist(), self.nonhp_bounds.tolist(), self.lengths.tolist()
        hp_dict, nonhp_dict = {}, {}
        for r, name in enumerate(names):
            if hb[r + 1] > hb[r]:
                hp_dict[name] = hp_pairs[hb[r]:hb[r + 1]]
                nonhp_dict[name] = non_pairs[nb[r]:nb[r + 1]]
            else:
                nonhp_dict[name] = [([(0, lens[r]), lens[r]])]
        return hp_dict, nonhp_dict


DOCUMENTS = ("hp_positions.json", "nonhp_positions.json")


def _pwrite_all(fd, data, offset):
    """``os.pwrite`` until every byte is on its way: one call may write less than it was given (a signal, a full pipe of the
    file system's, a quota edge); an unlooped call would leave a hole of NULs in the document and nobody would notice."""
    import errno
    import os
    view = memoryview(data)
    done = 0
    while done < len(view):
        n = os.pwrite(fd, view[done:], offset + done)
        if n <= 0:
            raise OSError(errno.EIO, "pwrite wrote nothing at offset %d" % (offset + done))
        done += n


def write_json_documents(directory, table, names, group=None):
    """Write the two chunk-coordinate documents of a (sharded) run, every rank its own part of them IN PARALLEL.

    Every rank of the job calls this with the ``ChunkTable`` of ITS reads and their names (ranks hold contiguous blocks of
    the file list, in rank order).  Each rank formats its members natively (``json_members``), the ranks exchange the byte
    counts (one small all-gather), rank 0 creates the two files at their final size with the braces in place, and then every
    rank ``pwrite``s its members -- preceded by ``", "`` when an earlier rank wrote any -- at its offset.  The files are byte
    for byte what one rank would have written (``{`` + members joined by ``", "`` + ``}``: ``json.dump`` of the reference's
    ``hp_dict`` / ``nonhp_dict``), and rank 0 does nothing that grows with the number of ranks: formatting 100 000 reads in
    one place costs ~0.1 s, more than eight MI355X need to classify them.  One node: all ranks must see ``directory`` as the same
    file system (the job shards the reads over the GPUs of ONE node; a shared file system extends it to several).

    Failure: the documents are built under ``<name>.part`` and renamed by rank 0 only after EVERY rank has reported its
    writes complete; creating, writing and publishing each end in ``sharding.agree_or_raise`` (which doubles as the barrier
    between the stages), so an I/O error on one rank -- EACCES, ENOSPC, a short write that will not complete -- raises on
    all of them at once, naming the rank, and leaves neither a document nor a ``.part`` behind (an exception aborts the run,
    as in the reference: catfish/split_f5.py:23-32).
    -> totals over all ranks: dict(reads, samples, reads_with_hp, hp_chunks, bytes) on every rank."""
    import os
    import torch.distributed as dist
    from .sharding import agree_or_raise
    texts = table.json_members(names)
    mine = (len(texts[0]), len(texts[1]), len(table), int(table.lengths.sum()), int(table.has_hp.sum()), int(table.hp_bounds[-1]))
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if distributed:
        rank = dist.get_rank()
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, mine, group=group)
    else:
        rank, every = 0, [mine]
    layout = []                                             # per document: (my offset, my separator, total size)
    for d in range(len(DOCUMENTS)):
        pos, seen, my_off, my_sep = 1, False, 0, b""
        for r, counts in enumerate(every):
            n = counts[d]
            sep = b", " if (n and seen) else b""
            if r == rank:
                my_off, my_sep = pos, sep
            pos += len(sep) + n
            seen = seen or n > 0
        layout.append((my_off, my_sep, pos + 1))
    finals = [os.path.join(directory, name) for name in DOCUMENTS]
    parts = [path + ".part" for path in finals]

    def stage(what, work, leftovers):
        """Run this rank's share of a stage, then let every rank learn how it went; on any failure rank 0 removes what exists."""
        error = None
        try:
            work()
        except Exception as exc:          # noqa: BLE001 -- ANY failure (MemoryError building the buffer, a TypeError out of pwrite), not
            error = exc                   # only OSError: a rank that skipped the agreement would leave its peers waiting in it
        try:
            agree_or_raise(error, what, group=group)
        except Exception:
            if rank == 0:
                for path in leftovers:
                    try:
                        os.unlink(path)
                    except OSError:
                        pass
            raise

    def create():
        if rank != 0:
            return
        for path, (_off, _sep, total) in zip(parts, layout):
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                os.ftruncate(fd, total)
                _pwrite_all(fd, b"{", 0)
                _pwrite_all(fd, b"}", total - 1)
            finally:
                os.close(fd)

    def write():
        for d, path in enumerate(parts):
            if texts[d]:
                fd = os.open(path, os.O_WRONLY)
                try:
                    _pwrite_all(fd, layout[d][1] + texts[d], layout[d][0])
                finally:
                    os.close(fd)

    def publish():
        if rank == 0:
            for part, final in zip(parts, finals):
                os.replace(part, final)

    stage("creating the chunk documents", create, parts)        # the files exist at their final size before anyone writes into them
    stage("writing the chunk documents", write, parts)
    stage("publishing the chunk documents", publish, parts + finals)   # complete documents under their names when any rank returns
    return {"reads": sum(c[2] for c in every), "samples": sum(c[3] for c in every), "reads_with_hp": sum(c[4] for c in every),
            "hp_chunks": sum(c[5] for c in every), "bytes": [total for _off, _sep, total in layout]}
