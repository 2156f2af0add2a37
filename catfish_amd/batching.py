"""Packed multi-read batches: many reads through ONE launch of the forward pass.

The reference runs one ``sess.run`` per read (catfish/catfish:55-56 -> infer.py:44).  Windows
are independent, so here all windows of a batch of reads -- of any mix of lengths -- are
packed into one ``[sum N_i, 35]`` tensor with a per-read offset table; no padding is added
beyond each read's own tail (infer.py:31-38).  Thresholding + ``correct_short`` run on the
device (cf_postprocess), span extraction (``hp_in_pred``) on the host over the uint8 labels.
"""
from __future__ import annotations

import numpy as np

from .infer import WINDOW_SIZE, padding_size_for

EXT_LEFT = 11    # hp_in_pred defaults, infer.py:141
EXT_RIGHT = 16


class quiet_gc(object):
    """Context manager: Python's cyclic garbage collector off while per-read span lists are built.

    The reference's result type is a list of ``[start, end]`` lists per read: ten thousand small container objects per
    batch.  Every 700 of them trigger a collection, and with torch imported a collection walks millions of live objects:
    measured on the MI355X box (tools/prof_shard.py) list building costs 2-6 ms per 256-read batch with the collector on
    and 0.3 ms with it off -- the difference between a host-bound and a GPU-bound pipeline (3.4 ms per batch).  Lists of
    ints hold no reference cycles, so reference counting frees them; the collector's previous state is restored on exit."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self._was:
            import gc
            gc.enable()
        return False


class PackedReads(object):
    """Normalised reads packed window-major.

    x            float32 [n_windows, 35]   zero padded per read (infer.py:31-43)
    win_offsets  int64 [n_reads + 1]       first window of every read
    lengths      int64 [n_reads]           real (un-padded) samples per read
    """

    def __init__(self, x, win_offsets, lengths):
        self.x = x
        self.win_offsets = win_offsets
        self.lengths = lengths

    @property
    def n_reads(self):
        return int(self.lengths.shape[0])

    @property
    def n_windows(self):
        return int(self.win_offsets[-1])

    @property
    def sample_offsets(self):
        return self.win_offsets * WINDOW_SIZE


def pack_reads(signals, window=WINDOW_SIZE):
    """signals: iterable of 1-D normalised arrays -> PackedReads."""
    signals = [np.asarray(s).reshape(-1) for s in signals]
    lengths = np.array([len(s) for s in signals], dtype=np.int64)
    pads = np.array([padding_size_for(int(n), window) for n in lengths], dtype=np.int64)
    n_win = (lengths + pads) // window
    win_offsets = np.zeros(len(signals) + 1, dtype=np.int64)
    np.cumsum(n_win, out=win_offsets[1:])
    x = np.zeros(int(win_offsets[-1]) * window, dtype=np.float32)
    for s, off in zip(signals, win_offsets[:-1] * window):
        x[off:off + len(s)] = s
    return PackedReads(x.reshape(-1, window), win_offsets, lengths)


def length_buckets(lengths, max_windows, window=WINDOW_SIZE):
    """Greedy buckets of read indices (longest first) holding at most ``max_windows`` windows each.

    Buckets only bound the launch size: windows are independent, so reads of different lengths
    share a bucket without any extra padding.
    """
    lengths = np.asarray(lengths, dtype=np.int64)
    n_win = np.array([(int(n) + padding_size_for(int(n), window)) // window for n in lengths], dtype=np.int64)
    order = np.argsort(-n_win, kind="stable")
    buckets, cur, cur_w = [], [], 0
    for i in order:
        w = int(n_win[i])
        if cur and cur_w + w > max_windows:
            buckets.append(cur)
            cur, cur_w = [], 0
        cur.append(int(i))
        cur_w += w
    if cur:
        buckets.append(cur)
    return buckets


def spans_from_labels(labels, sample_offsets, n_reads, ext_left=EXT_LEFT, ext_right=EXT_RIGHT):
    """Packed corrected labels (padding = 0) -> per-read lists of [start - 11, end + 16] (hp_in_pred).

    Every read's padding is at least one zero label, so positive runs never span two reads and
    one global run-length pass serves all reads.
    """
    lab = np.asarray(labels).astype(np.int8)
    d = np.diff(np.concatenate(([0], lab, [0])))
    starts = np.flatnonzero(d == 1)
    ends = np.flatnonzero(d == -1)
    read_of = np.searchsorted(sample_offsets, starts, side="right") - 1
    base = sample_offsets[read_of]
    out = [[] for _ in range(n_reads)]
    for r, s, e in zip(read_of.tolist(), (starts - base).tolist(), (ends - base).tolist()):
        out[r].append([s - ext_left, e + ext_right])
    return out


def spans_from_runs(starts, ends, sample_offsets, n_reads, ext_left=EXT_LEFT, ext_right=EXT_RIGHT):
    """Sorted packed run boundaries (engine.spans_device) -> per-read [start - 11, end + 16] lists.

    Vectorised: one ``tolist`` of the whole span table, then one slice per read (runs are sorted, so a read's
    spans are contiguous)."""
    starts = np.asarray(starts, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    sample_offsets = np.asarray(sample_offsets, dtype=np.int64)
    read_of = np.searchsorted(sample_offsets, starts, side="right") - 1
    base = sample_offsets[read_of]
    pairs = np.stack([starts - base - ext_left, ends - base + ext_right], axis=1).tolist()
    bounds = np.concatenate(([0], np.cumsum(np.bincount(read_of, minlength=n_reads)[:n_reads]))).tolist()
    return [pairs[bounds[r]:bounds[r + 1]] for r in range(n_reads)]


def infer_packed(engine, packed, threshold=0.5, min_run=15, return_probs=False):
    """PackedReads -> list of (spans, read length) per read, optionally with per-read probabilities."""
    import torch
    dev = torch.device("cuda", engine.device)
    x = torch.from_numpy(packed.x).to(dev, non_blocking=True)
    offs = torch.from_numpy(packed.sample_offsets).to(dev)
    lens = torch.from_numpy(packed.lengths).to(dev)
    probs = engine.infer_device(x)
    labels = engine.postprocess_device(probs, offs, lens, threshold=threshold, min_run=min_run)
    starts, ends = engine.spans_device(labels)
    spans = spans_from_runs(starts, ends, packed.sample_offsets, packed.n_reads)
    result = [(spans[i], int(packed.lengths[i])) for i in range(packed.n_reads)]
    if return_probs:
        p = probs.cpu().numpy()
        so = packed.sample_offsets
        return result, [p[so[i]:so[i] + packed.lengths[i]] for i in range(packed.n_reads)]
    return result


def infer_reads(model, signals, max_windows=None, threshold=0.5, min_run=15):
    """Many normalised reads -> [(spans, length)] in input order, length-bucketed packed launches."""
    engine = model.engine if hasattr(model, "engine") else model
    if engine is None:
        raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
    signals = [np.asarray(s).reshape(-1) for s in signals]
    if max_windows is None:
        max_windows = 32768
    out = [None] * len(signals)
    with quiet_gc():
        for bucket in length_buckets([len(s) for s in signals], max_windows):
            packed = pack_reads([signals[i] for i in bucket])
            for i, res in zip(bucket, infer_packed(engine, packed, threshold, min_run)):
                out[i] = res
    return out


def infer_reads_dac(model, dac_reads, max_windows=None, threshold=0.5, min_run=15, return_probs=False):
    """Raw int16 DAC reads (leader already trimmed) -> [(spans, length)] with normalisation ON DEVICE.

    Uploads 2 B per sample; median/MAD normalisation, padding and window packing run in
    ``cf_normalize`` (bit-identical to infer.normalize_raw_signal cast to float32), then the
    forward pass and the device post-processing as in ``infer_packed``.
    """
    import torch
    engine = model.engine if hasattr(model, "engine") else model
    if engine is None:
        raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
    dac_reads = [np.ascontiguousarray(np.asarray(r).reshape(-1), dtype=np.int16) for r in dac_reads]
    if max_windows is None:
        max_windows = 32768
    dev = torch.device("cuda", engine.device)
    out = [None] * len(dac_reads)
    probs_out = [None] * len(dac_reads)
    for bucket in length_buckets([len(r) for r in dac_reads], max_windows):
        lengths = np.array([len(dac_reads[i]) for i in bucket], dtype=np.int64)
        dac_off = np.zeros(len(bucket) + 1, dtype=np.int64)
        np.cumsum(lengths, out=dac_off[1:])
        n_win = np.array([(int(n) + padding_size_for(int(n))) // WINDOW_SIZE for n in lengths], dtype=np.int64)
        win_off = np.zeros(len(bucket) + 1, dtype=np.int64)
        np.cumsum(n_win, out=win_off[1:])
        flat = np.concatenate([dac_reads[i] for i in bucket]) if len(bucket) else np.zeros(0, np.int16)
        d_dac = torch.from_numpy(flat).to(dev)
        d_doff = torch.from_numpy(dac_off).to(dev)
        d_woff = torch.from_numpy(win_off).to(dev)
        x = torch.empty(int(win_off[-1]), WINDOW_SIZE, dtype=torch.float32, device=dev)
        engine.normalize_device(d_dac, d_doff, d_woff, out=x)
        probs = engine.infer_device(x)
        s_off = win_off * WINDOW_SIZE
        labels = engine.postprocess_device(probs, torch.from_numpy(s_off).to(dev), torch.from_numpy(lengths).to(dev),
                                           threshold=threshold, min_run=min_run)
        starts, ends = engine.spans_device(labels)
        spans = spans_from_runs(starts, ends, s_off, len(bucket))
        p_host = probs.cpu().numpy() if return_probs else None
        for k, i in enumerate(bucket):
            out[i] = (spans[k], int(lengths[k]))
            if return_probs:
                probs_out[i] = p_host[s_off[k]:s_off[k] + lengths[k]]
    return (out, probs_out) if return_probs else out
