"""Host-to-host streaming pipeline: raw int16 DAC reads in, homopolymer spans out.

One batch = many reads packed back to back.  Per batch the device runs
``cf_normalize -> cf_infer -> cf_postprocess -> cf_spans``; only 2 B/sample go up and two short
run-boundary lists come down.  Batches are double-buffered: the H2D copy of batch k+1 (copy stream,
pinned staging) overlaps the kernels of batch k (compute stream); the run lists of batch k go down on a
third stream into pinned buffers as soon as its cf_spans is done (never queued behind batch k+1's kernels),
and the host-side span assembly of batch k-1 overlaps all of it.  This is the "secondary" (PCIe-inclusive) rate of SURVEY.md 8d; it is never
bench.py's ``value``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as N
from .batching import spans_from_runs
from .infer import WINDOW_SIZE


class _Ticket(object):
    __slots__ = ("lengths", "s_off", "starts", "ends", "counts", "starts_h", "ends_h", "counts_h", "done", "max_runs", "labels",
                 "keep")


class ReadPipeline(object):
    def __init__(self, engine, max_samples_per_batch, threshold=0.5, min_run=15, depth=2, overlap_kernels=None):
        import torch
        self.torch = torch
        self.eng = engine
        self.dev = torch.device("cuda", engine.device)
        self.threshold = float(threshold)
        self.min_run = int(min_run)
        # (All three streams at one priority: raising the forward pass's -- tried in round 5 -- starves the next batch's ingest until
        # the running biGRU launch ends and serialises the two; the CLI's 131 072-window batches lost 8 %,
        # profiles/r05_ab_pipeline_knobs.log.)
        self.compute = torch.cuda.Stream(self.dev)
        self.copy = torch.cuda.Stream(self.dev)
        self.down = torch.cuda.Stream(self.dev)
        # ``depth`` batches may be in flight (staged, launched, not yet collected).  Two is the measured optimum: with three the
        # extra batch's side-stream kernels (normalisation, post-processing) crowd the forward pass of the batch in between
        # (fp32 3.50 -> 3.68 ms per batch, bf16 1.00 -> 1.44 ms, tools/bench_e2e.py); short bf16 batches are better served
        # by more reads per batch (1024 reads: 1.55 G samples/s host to host).  Re-measured in round 5 with the 5x cheaper side kernels
        # (tools/bench_e2e.py --depth 3): fp32 equal (309.8 vs 310.4 M samples/s), bf16 still 12 % slower (1.26 vs 1.44 G)
        self.depth = max(2, int(depth))
        # where the small kernels run: True = normalisation on the copy stream and post-processing on the download stream, beside
        # the neighbouring batches' forward passes; False = all kernels of a batch in order on the compute stream (only the
        # H2D / D2H copies overlap).  CATFISH_PIPE_OVERLAP=0/1 overrides (A/B knob for tools/).
        if overlap_kernels is None and "CATFISH_PIPE_OVERLAP" in os.environ:
            overlap_kernels = os.environ["CATFISH_PIPE_OVERLAP"] != "0"
        self.overlap_kernels = True if overlap_kernels is None else bool(overlap_kernels)
        self.out = [None] * self.depth                           # pinned (starts, ends, counts) per in-flight slot, grown on demand
        self.inflight = [None] * self.depth
        self.cap = int(max_samples_per_batch)
        # two pinned staging buffers (double buffering)
        self.stage = [torch.empty(self.cap, dtype=torch.int16, pin_memory=True) for _ in range(self.depth)]
        self.stage_free = [torch.cuda.Event() for _ in range(self.depth)]
        self.tab = [None] * self.depth                           # pinned offset / length tables per staging slot
        self.k = 0
        n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
        self._file_threads = max(2, min(8, n_cpu // 4))

    def close(self):
        """Stop the helper thread of ``preload_listing`` (if one was started); the pipeline must not be used afterwards."""
        loader, self._loader = getattr(self, "_loader", None), None
        if loader is not None:
            loader.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _claim_slot(self):
        """The staging slot of the next batch, once the H2D copy that last read from it is done."""
        slot = self.k % self.depth
        if self.inflight[slot] is not None:
            raise RuntimeError("ReadPipeline: at most %d batches in flight; collect() the oldest ticket first" % self.depth)
        self.stage_free[slot].synchronize()                     # previous H2D out of this staging buffer is done
        return slot

    def submit_files(self, paths, n_threads=None):
        """Launch one batch straight from files: one-dimensional little-endian int16 ``.npy`` reads (the format
        ``infer.load_dac`` takes when there is no HDF5) are read by the library's host thread pool (``cf_load_npy_int16``)
        directly into the pinned staging slot -- no per-file Python, no intermediate arrays.  Returns a ticket, or None when
        this batch needs the general loader (a file of another kind, or reads that do not fit the staging buffer): nothing
        has been consumed then."""
        paths = [os.fsencode(p) for p in paths]
        if not paths or not all(p.endswith(b".npy") for p in paths):      # infer.load_dac's own dispatch: only .npy takes this path
            return None
        if n_threads is None:
            # a quarter of the CPUs this rank owns (placement.bind: 32 per rank on an 8-GPU node), 2 .. 8: a 1110-file batch takes
            # 6.8 / 2.3 / 1.8 / 1.7 ms with 1 / 4 / 8 / 16 threads (tools/exp_loader_threads.py) -- past 8 nothing is gained, and in
            # bf16 the batch's forward pass takes 2.3 ms, so the loader sets the pace below that
            n_threads = self._file_threads
        slot = self._claim_slot()
        blob = b"\x00".join(paths) + b"\x00"
        bounds = np.zeros(len(paths) + 1, dtype=np.int64)
        np.cumsum([len(p) + 1 for p in paths], out=bounds[1:])
        lengths = np.empty(len(paths), dtype=np.int64)
        total = C.c_int64(0)
        rc = self.eng._lib.cf_load_npy_int16(blob, bounds.ctypes.data_as(C.c_void_p), len(paths),
                                             C.c_void_p(self.stage[slot].data_ptr()), self.cap,
                                             lengths.ctypes.data_as(C.c_void_p), C.byref(total), int(n_threads))
        if rc == N.CF_ERR_NOMEM:
            N.check(rc)         # out of host memory is not "a file of another kind": loading the same bytes again through the general loader would only hide it
        if rc != N.CF_OK:
            return None
        self.k += 1
        return self._launch(slot, lengths, int(total.value))

    def submit_listing(self, listing, lo, hi, n_threads=None):
        """``submit_files`` for entries [lo, hi) of a ``sharding.DirListing``: the library opens them relative to the listing's
        directory (``cf_listing_load_npy_int16``), so not even a path string per file is built -- 12 500 files cost a rank 14 ms of
        Python that way, a third of a bf16 shard's time.  None when the batch needs the general loader (an entry that is not a
        ``.npy`` int16 vector, or reads that do not fit the staging buffer): nothing has been consumed then."""
        if hi <= lo:
            return None
        slot = self._claim_slot()
        lengths = np.empty(hi - lo, dtype=np.int64)
        total = C.c_int64(0)
        rc = self.eng._lib.cf_listing_load_npy_int16(listing._handle, int(lo), int(hi), C.c_void_p(self.stage[slot].data_ptr()), self.cap,
                                                     lengths.ctypes.data_as(C.c_void_p), C.byref(total),
                                                     int(self._file_threads if n_threads is None else n_threads))
        if rc == N.CF_ERR_NOMEM:
            N.check(rc)
        if rc != N.CF_OK:
            return None
        self.k += 1
        return self._launch(slot, lengths, int(total.value))

    # ---- the same with the file reads one batch AHEAD, in a helper thread (the native call releases the GIL): while the main thread
    # waits for the GPU and assembles the results of the batch before, the next batch's files are read into the other staging slot.
    # In bf16 a 1110-read batch takes the GPU 2.3 ms and its files 2.1 ms to read: back to back on one thread the host set the pace.
    def preload_listing(self, listing, lo, hi, n_threads=None):
        """Start reading entries [lo, hi) of ``listing`` into the staging slot the NEXT launch will use; -> a handle for
        ``launch_preloaded``.  At most one preload may be pending, and nothing else may be submitted until it has been launched or
        dropped (``drop_preloaded``): the caller is ``EngineBatchRunner.run_listing``."""
        import concurrent.futures
        if getattr(self, "_loader", None) is None:
            self._loader = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="catfish-files")
        slot = self.k % self.depth
        stage, cap = self.stage[slot], self.cap
        threads = int(self._file_threads if n_threads is None else n_threads)
        lib, handle, free = self.eng._lib, listing._handle, self.stage_free[slot]

        def work():
            free.synchronize()                                   # the H2D copy that last read from this staging buffer is done
            lengths = np.empty(max(0, hi - lo), dtype=np.int64)
            total = C.c_int64(0)
            rc = lib.cf_listing_load_npy_int16(handle, int(lo), int(hi), C.c_void_p(stage.data_ptr()), cap,
                                               lengths.ctypes.data_as(C.c_void_p), C.byref(total), threads)
            # the library's error text is per THREAD: read it here, where the call ran (the main thread would see a stale or empty one)
            msg = lib.cf_last_error().decode("utf-8", "replace") if rc != N.CF_OK else ""
            return rc, lengths, int(total.value), msg
        return (slot, self._loader.submit(work))

    def launch_preloaded(self, pre):
        """Launch the batch ``preload_listing`` read.  None when it needs the general loader (nothing has been consumed)."""
        slot, fut = pre
        rc, lengths, total, msg = fut.result()
        if rc == N.CF_ERR_NOMEM:
            raise N.CatfishHipError(rc, msg)
        self.last_preload_note = msg if rc != N.CF_OK else ""    # why the batch goes to the general loader (names the offending file)
        if rc != N.CF_OK or len(lengths) == 0:
            return None
        if slot != self.k % self.depth or self.inflight[slot] is not None:
            raise RuntimeError("ReadPipeline: a preloaded batch must be launched before anything else is submitted")
        self.k += 1
        return self._launch(slot, lengths, total)

    @staticmethod
    def drop_preloaded(pre):
        """Wait for a preload that will not be launched (its staging slot is reusable afterwards)."""
        if pre is not None:
            pre[1].result()

    def submit(self, dac_reads):
        """Launch one batch asynchronously; returns a ticket for ``collect``."""
        torch = self.torch
        lengths = np.array([len(r) for r in dac_reads], dtype=np.int64)
        total = int(lengths.sum())
        if total > self.cap:
            if len(dac_reads) > 1:
                raise ValueError("batch of %d samples exceeds max_samples_per_batch=%d" % (total, self.cap))
            # a single read longer than the batch size: grow the pinned staging buffers once
            for ev in self.stage_free:
                ev.synchronize()
            self.cap = total
            self.stage = [torch.empty(self.cap, dtype=torch.int16, pin_memory=True) for _ in range(self.depth)]
        slot = self._claim_slot()
        self.k += 1
        host = self.stage[slot].numpy()
        if len(dac_reads) == 1:
            host[:total] = dac_reads[0]
        elif total:
            np.concatenate(dac_reads, out=host[:total], casting="unsafe")     # one C call instead of a Python loop over the reads
        return self._launch(slot, lengths, total)

    def _launch(self, slot, lengths, total):
        """Everything after staging: tables up, H2D, normalise, forward, post-process, run lists down."""
        torch = self.torch
        n_reads = len(lengths)
        dac_off = np.zeros(n_reads + 1, dtype=np.int64)
        np.cumsum(lengths, out=dac_off[1:])
        n_win = lengths // WINDOW_SIZE + 1                      # infer.py:32-36: a multiple of 35 gets a full extra window
        win_off = np.zeros(n_reads + 1, dtype=np.int64)
        np.cumsum(n_win, out=win_off[1:])
        # the four small tables go up from PINNED memory too, as one copy: a pageable source makes the runtime pin user
        # pages for the transfer, and host allocator activity (munmap) near such mappings stalls the GPU queues
        n_r = n_reads
        n_tab = 4 * (n_r + 1)
        if self.tab[slot] is None or self.tab[slot].numel() < n_tab:
            self.tab[slot] = torch.empty(max(n_tab, 1024), dtype=torch.int64, pin_memory=True)
        tab = self.tab[slot].numpy()
        tab[0:n_r + 1] = dac_off
        tab[n_r + 1:2 * n_r + 2] = win_off
        tab[2 * n_r + 2:3 * n_r + 3] = win_off * WINDOW_SIZE
        tab[3 * n_r + 3:4 * n_r + 3] = lengths
        with torch.cuda.stream(self.copy):
            d_dac = self.stage[slot][:total].to(self.dev, non_blocking=True)
            d_tab = self.tab[slot][:n_tab].to(self.dev, non_blocking=True)
            d_doff, d_woff = d_tab[0:n_r + 1], d_tab[n_r + 1:2 * n_r + 2]
            d_soff, d_len = d_tab[2 * n_r + 2:3 * n_r + 3], d_tab[3 * n_r + 3:4 * n_r + 3]
            self.stage_free[slot].record(self.copy)
            n_windows = int(win_off[-1])
            h2d_done = torch.cuda.Event()
            h2d_done.record(self.copy)
        k_norm = self.copy if self.overlap_kernels else self.compute
        k_post = self.down if self.overlap_kernels else self.compute
        with torch.cuda.stream(k_norm):
            # overlap mode: normalisation rides on the copy stream and overlaps the previous batch's biGRU kernels (which leave
            # wave slots and 15 KiB of LDS free on every CU) instead of delaying this batch's
            k_norm.wait_event(h2d_done)
            x = torch.empty(n_windows, WINDOW_SIZE, dtype=torch.float32, device=self.dev)
            self.eng.normalize_device(d_dac, d_doff, d_woff, out=x, stream=k_norm)
            copied = torch.cuda.Event()
            copied.record(k_norm)
        t = _Ticket()
        with torch.cuda.stream(self.compute):                    # overlap mode: the compute stream only ever holds the forward pass
            self.compute.wait_event(copied)
            probs = self.eng.infer_device(x, stream=self.compute)
            infer_done = torch.cuda.Event()
            infer_done.record(self.compute)
        with torch.cuda.stream(k_post):                          # overlap mode: post-processing overlaps the next batch's forward pass
            k_post.wait_event(infer_done)
            # threshold + correct_short + run boundaries in ONE launch, straight from the probabilities; the labels themselves are
            # never written (cf_postprocess_spans with labels = NULL) -- except for min_run > 64, which takes the two older kernels
            max_runs = n_windows * WINDOW_SIZE // self.min_run + 16
            total = n_windows * WINDOW_SIZE
            labels = torch.empty(total, dtype=torch.uint8, device=self.dev) if self.min_run > 64 else None
            t.starts = torch.empty(max_runs, dtype=torch.int64, device=self.dev)
            t.ends = torch.empty(max_runs, dtype=torch.int64, device=self.dev)
            t.counts = torch.empty(2, dtype=torch.int64, device=self.dev)
            N.check(self.eng._lib.cf_postprocess_spans(
                self.eng._handle, C.c_void_p(probs.data_ptr()), C.c_void_p(d_soff.data_ptr()), C.c_void_p(d_len.data_ptr()), n_reads, total,
                float(self.threshold), int(self.min_run), C.c_void_p(labels.data_ptr()) if labels is not None else None, max_runs,
                C.c_void_p(t.starts.data_ptr()), C.c_void_p(t.ends.data_ptr()), C.c_void_p(t.counts.data_ptr()),
                C.c_void_p(k_post.cuda_stream)))
            spans_done = torch.cuda.Event()
            spans_done.record(k_post)
        if self.out[slot] is None or self.out[slot][0].numel() < max_runs:
            self.out[slot] = (torch.empty(max_runs, dtype=torch.int64, pin_memory=True),
                              torch.empty(max_runs, dtype=torch.int64, pin_memory=True),
                              torch.empty(2, dtype=torch.int64, pin_memory=True))
        t.starts_h, t.ends_h, t.counts_h = self.out[slot]
        with torch.cuda.stream(self.down):                       # D2H of the (unsorted) run lists, whole capacity: ~1 MB
            self.down.wait_event(spans_done)
            t.starts_h[:max_runs].copy_(t.starts, non_blocking=True)
            t.ends_h[:max_runs].copy_(t.ends, non_blocking=True)
            t.counts_h.copy_(t.counts, non_blocking=True)
            t.done = torch.cuda.Event()
            t.done.record(self.down)
        t.lengths, t.s_off, t.max_runs, t.labels = lengths, win_off * WINDOW_SIZE, max_runs, labels
        t.keep = (d_dac, d_tab, x, probs, slot)                  # keep device buffers alive until collected
        self.inflight[slot] = t
        return t

    def collect(self, t, as_lists=True):
        """Wait for a batch and assemble [(spans, read length)] in input order.

        ``as_lists=False`` skips the per-read Python lists and returns the span table as arrays
        ``(read_index, start - 11, end + 16, read_lengths)`` (what a high-rate consumer wants)."""
        t.done.synchronize()
        try:
            self.eng.check_error()                               # asynchronous launches of this batch (a device-side error is
            n_s, n_e = (int(v) for v in t.counts_h.tolist())     # sticky until engine.clear_error())
            if n_s != n_e or n_s > t.max_runs:
                raise RuntimeError("cf_spans returned %d starts / %d ends (capacity %d)" % (n_s, n_e, t.max_runs))
            starts = np.sort(t.starts_h[:n_s].numpy())           # np.sort copies out of the pinned slot
            ends = np.sort(t.ends_h[:n_e].numpy())
        finally:
            # the slot is free again and the batch's device buffers are released whether or not its results were good: a
            # failed batch must not leave the pipeline claiming "too many batches in flight" on the next submit
            self.inflight[t.keep[-1]] = None
            t.keep = None
            t.starts = t.ends = t.counts = None
        if not as_lists:
            read_of = np.searchsorted(t.s_off, starts, side="right") - 1
            base = t.s_off[read_of]
            return read_of, starts - base - 11, ends - base + 16, t.lengths
        spans = spans_from_runs(starts, ends, t.s_off, len(t.lengths))
        return [(spans[i], int(t.lengths[i])) for i in range(len(t.lengths))]

    def run(self, batches, as_lists=True):
        """Iterate over batches (lists of int16 reads) with up to ``depth - 1`` batches in flight ahead; yields results in order."""
        from collections import deque
        from .batching import quiet_gc
        pending = deque()
        with quiet_gc():                      # list building must not trigger collections that walk the whole process
            for b in batches:
                if len(pending) == self.depth:
                    yield self.collect(pending.popleft(), as_lists)
                pending.append(self.submit(b))
                if len(pending) == self.depth:
                    yield self.collect(pending.popleft(), as_lists)
            while pending:
                yield self.collect(pending.popleft(), as_lists)
