"""Thin host layer over the C ABI: owns one cf_model on one MI355X.

PyTorch is used only as plumbing (device buffers, the current HIP stream);
numpy callers go through ``cf_infer_host`` and need no torch at all.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

WINDOW = N.CF_WINDOW
DEFAULT_MAX_WINDOWS = 32768   # one pass covers 256 reads of 4096 samples (30 208 windows)


def device_identity(device):
    """-> (pci bus id ``"dddd:bb:dd.f"``, uuid as 32 hex digits) of HIP device ``device`` of this process (``cf_device_identity``):
    which card a rank drives, for ``placement.verify`` and the benchmark line.  Opens the HIP runtime."""
    bus, uuid = C.create_string_buffer(64), C.create_string_buffer(33)
    N.check(N.lib().cf_device_identity(int(device), bus, 64, uuid, 33))
    return bus.value.decode("ascii", "replace").lower(), uuid.value.decode("ascii", "replace")


class HipEngine(object):
    """The forward pass of one ResNetRNN checkpoint on one GPU."""

    def __init__(self, weights, layer_size=64, n_layers=3, layer_size_res=32, n_layers_res=2,
                 device=0, max_windows_per_pass=DEFAULT_MAX_WINDOWS, bn_epsilon=1e-3, n_streams=0,
                 precision="fp32", fuse_layers=None):
        self._lib = N.lib()
        self._handle = C.c_void_p()
        if precision not in N.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(N.PRECISIONS))
        self.precision = precision
        hp = N.cf_hparams(int(layer_size), int(n_layers), int(layer_size_res), int(n_layers_res),
                          WINDOW, float(bn_epsilon), int(max_windows_per_pass), int(n_streams),
                          N.PRECISIONS[precision], 0 if fuse_layers is None else (1 if fuse_layers else -1))
        w, keep = N.build_weight_structs(weights, int(n_layers), int(n_layers_res))
        N.check(self._lib.cf_model_create(C.byref(w), C.byref(hp), int(device), C.byref(self._handle)))
        del keep  # the library copied/re-tiled everything
        self.device = int(device)
        self.n_layers = int(n_layers)
        self.n_layers_res = int(n_layers_res)
        self.layer_size = int(layer_size)
        self.layer_size_res = int(layer_size_res)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.cf_model_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_identity(self):
        """(pci bus id, uuid hex) of the card this engine drives."""
        return device_identity(self.device)

    @property
    def workspace_bytes(self):
        return int(self._lib.cf_workspace_bytes(self._handle))

    # ------------------------------------------------------------------ inference
    @staticmethod
    def _check_windows(shape):
        if len(shape) == 3 and shape[2] == 1:
            shape = shape[:2]
        if len(shape) != 2 or shape[1] != WINDOW:
            raise ValueError("input must be [n_windows, %d(, 1)], got %s" % (WINDOW, tuple(shape)))
        return int(shape[0])

    def infer_host(self, x, return_logits=False):
        """numpy [N,35(,1)] any float dtype -> numpy float32 [N*35] (H2D/D2H inside).

        ``return_logits=True`` -> (probs, logits): the pre-sigmoid values of rnn_class.py:178-183."""
        x = np.asarray(x)
        n = self._check_windows(x.shape)
        x32 = np.ascontiguousarray(x.reshape(n, WINDOW), dtype=np.float32)
        out = np.empty(n * WINDOW, dtype=np.float32)
        if return_logits:
            logits = np.empty(n * WINDOW, dtype=np.float32)
            N.check(self._lib.cf_infer_host_logits(self._handle, x32.ctypes.data_as(C.c_void_p), n,
                                                   out.ctypes.data_as(C.c_void_p), logits.ctypes.data_as(C.c_void_p)))
            return out, logits
        N.check(self._lib.cf_infer_host(self._handle, x32.ctypes.data_as(C.c_void_p), n,
                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def check_error(self):
        """Raise if an earlier asynchronous launch of this model reported a device-side error (call after a sync)."""
        N.check(self._lib.cf_check_error(self._handle))

    def clear_error(self):
        """Reset the sticky device-side error after the results of the failed launch have been dropped (``cf_clear_error``):
        the engine is usable again."""
        N.check(self._lib.cf_clear_error(self._handle))

    def launch_regimes(self):
        """Switch points of the launcher (windows): dict(n_cu, hoist_max, coop_max, fuse_auto_min)."""
        out = (C.c_int64 * 4)()
        N.check(self._lib.cf_launch_regimes(self._handle, out))
        return dict(n_cu=int(out[0]), hoist_max=int(out[1]), coop_max=int(out[2]), fuse_auto_min=int(out[3]))

    def infer_device(self, x, out=None, stream=None, logits=None):
        """torch CUDA float32 tensor [N,35(,1)] -> torch CUDA float32 [N*35], async on the stream.

        ``logits``: optional float32 CUDA tensor of N*35 elements that receives the pre-sigmoid values."""
        import torch
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("infer_device needs a float32 CUDA tensor")
        if x.device.index != self.device:
            raise ValueError("tensor lives on cuda:%s, model on cuda:%d" % (x.device.index, self.device))
        n = self._check_windows(tuple(x.shape))
        x = x.contiguous()
        if out is None:
            out = torch.empty(n * WINDOW, dtype=torch.float32, device=x.device)
        elif out.numel() != n * WINDOW or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 tensor of %d elements" % (n * WINDOW))
        if stream is None:
            stream = torch.cuda.current_stream(x.device)
        if logits is not None:
            if logits.numel() != n * WINDOW or logits.dtype != torch.float32 or not logits.is_contiguous() or not logits.is_cuda:
                raise ValueError("logits must be a contiguous float32 CUDA tensor of %d elements" % (n * WINDOW))
            N.check(self._lib.cf_infer_logits(self._handle, C.c_void_p(x.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                              C.c_void_p(logits.data_ptr()), C.c_void_p(stream.cuda_stream)))
            return out
        N.check(self._lib.cf_infer(self._handle, C.c_void_p(x.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(stream.cuda_stream)))
        return out

    def postprocess_device(self, probs, read_offsets, read_lengths, threshold=0.5, min_run=15, out=None, stream=None):
        """Device threshold + correct_short over packed padded reads.

        probs: float32 CUDA [total]; read_offsets: int64 CUDA [n_reads+1] (padded sample offsets);
        read_lengths: int64 CUDA [n_reads] (real lengths) -> uint8 CUDA labels [total] (padding = 0).
        """
        import torch
        if not probs.is_cuda or probs.dtype != torch.float32 or not probs.is_contiguous():
            raise ValueError("probs must be a contiguous float32 CUDA tensor")
        for t in (read_offsets, read_lengths):
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("read_offsets/read_lengths must be contiguous int64 CUDA tensors")
        n_reads = int(read_lengths.numel())
        if int(read_offsets.numel()) != n_reads + 1:
            raise ValueError("read_offsets must have n_reads + 1 entries")
        total = int(probs.numel())
        if out is None:
            out = torch.empty(total, dtype=torch.uint8, device=probs.device)
        if stream is None:
            stream = torch.cuda.current_stream(probs.device)
        N.check(self._lib.cf_postprocess(self._handle, C.c_void_p(probs.data_ptr()),
                                         C.c_void_p(read_offsets.data_ptr()), C.c_void_p(read_lengths.data_ptr()),
                                         n_reads, total, float(threshold), int(min_run),
                                         C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream)))
        return out

    def postprocess_spans_device(self, probs, read_offsets, read_lengths, threshold=0.5, min_run=15, max_runs=None, labels=False,
                                 stream=None):
        """``postprocess_device`` + ``spans_device`` as ONE launch (``cf_postprocess_spans``) -> (starts, ends) numpy int64, sorted
        ascending (packed positions; ends exclusive), and the uint8 CUDA labels as a third value when ``labels=True``."""
        import torch
        if not probs.is_cuda or probs.dtype != torch.float32 or not probs.is_contiguous():
            raise ValueError("probs must be a contiguous float32 CUDA tensor")
        for t in (read_offsets, read_lengths):
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("read_offsets/read_lengths must be contiguous int64 CUDA tensors")
        n_reads, total = int(read_lengths.numel()), int(probs.numel())
        if int(read_offsets.numel()) != n_reads + 1:
            raise ValueError("read_offsets must have n_reads + 1 entries")
        if max_runs is None:
            max_runs = total // max(1, int(min_run)) + 16
        dev = probs.device
        lab = torch.empty(total, dtype=torch.uint8, device=dev) if labels else None      # NULL: the library never writes labels
        starts = torch.empty(max_runs, dtype=torch.int64, device=dev)
        ends = torch.empty(max_runs, dtype=torch.int64, device=dev)
        counts = torch.empty(2, dtype=torch.int64, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        N.check(self._lib.cf_postprocess_spans(self._handle, C.c_void_p(probs.data_ptr()), C.c_void_p(read_offsets.data_ptr()),
                                               C.c_void_p(read_lengths.data_ptr()), n_reads, total, float(threshold), int(min_run),
                                               C.c_void_p(lab.data_ptr()) if lab is not None else None, int(max_runs),
                                               C.c_void_p(starts.data_ptr()), C.c_void_p(ends.data_ptr()), C.c_void_p(counts.data_ptr()),
                                               C.c_void_p(stream.cuda_stream)))
        n_s, n_e = (int(v) for v in counts.cpu().tolist())      # synchronises the stream
        self.check_error()
        if n_s != n_e:
            raise RuntimeError("cf_postprocess_spans: %d run starts but %d run ends" % (n_s, n_e))
        if n_s > max_runs:
            return self.postprocess_spans_device(probs, read_offsets, read_lengths, threshold, min_run, n_s, labels, stream)
        out = (np.sort(starts[:n_s].cpu().numpy()), np.sort(ends[:n_e].cpu().numpy()))
        return out + (lab,) if labels else out

    def spans_device(self, labels, max_runs=None, stream=None):
        """Device run-length pass over corrected labels -> (starts, ends) numpy int64, sorted ascending
        (packed positions; ends exclusive).  Only the two short lists cross PCIe."""
        import torch
        if labels.dtype != torch.uint8 or not labels.is_cuda or not labels.is_contiguous():
            raise ValueError("labels must be a contiguous uint8 CUDA tensor")
        total = int(labels.numel())
        if max_runs is None:
            max_runs = total // 15 + 16          # correct_short leaves runs of >= 15 samples
        dev = labels.device
        starts = torch.empty(max_runs, dtype=torch.int64, device=dev)
        ends = torch.empty(max_runs, dtype=torch.int64, device=dev)
        counts = torch.empty(2, dtype=torch.int64, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        N.check(self._lib.cf_spans(self._handle, C.c_void_p(labels.data_ptr()), total, int(max_runs),
                                   C.c_void_p(starts.data_ptr()), C.c_void_p(ends.data_ptr()),
                                   C.c_void_p(counts.data_ptr()), C.c_void_p(stream.cuda_stream)))
        n_s, n_e = (int(v) for v in counts.cpu().tolist())      # synchronises the stream
        self.check_error()                                        # the labels came from asynchronous launches
        if n_s != n_e:
            raise RuntimeError("cf_spans: %d run starts but %d run ends" % (n_s, n_e))
        if n_s > max_runs:
            return self.spans_device(labels, max_runs=n_s, stream=stream)
        return np.sort(starts[:n_s].cpu().numpy()), np.sort(ends[:n_e].cpu().numpy())

    def normalize_device(self, dac, dac_offsets, win_offsets, out=None, stream=None):
        """Device median/MAD normalisation + padding + window packing of many int16 reads.

        dac: int16 CUDA [total samples]; dac_offsets / win_offsets: int64 CUDA [n_reads + 1]
        -> float32 CUDA [n_windows, 35] (n_windows = win_offsets[-1], given by ``out`` or computed).
        """
        import torch
        if dac.dtype != torch.int16 or not dac.is_cuda or not dac.is_contiguous():
            raise ValueError("dac must be a contiguous int16 CUDA tensor")
        for t in (dac_offsets, win_offsets):
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("offset tables must be contiguous int64 CUDA tensors")
        n_reads = int(dac_offsets.numel()) - 1
        if int(win_offsets.numel()) != n_reads + 1:
            raise ValueError("dac_offsets and win_offsets must have the same length")
        if out is None:
            out = torch.empty(int(win_offsets[-1].item()), WINDOW, dtype=torch.float32, device=dac.device)
        if stream is None:
            stream = torch.cuda.current_stream(dac.device)
        N.check(self._lib.cf_normalize(self._handle, C.c_void_p(dac.data_ptr()), C.c_void_p(dac_offsets.data_ptr()),
                                       C.c_void_p(win_offsets.data_ptr()), n_reads, C.c_void_p(out.data_ptr()),
                                       C.c_void_p(stream.cuda_stream)))
        return out

    # ------------------------------------------------------------------ profiling / debug
    def profile_enable(self, on=True, every=1):
        """Per-kernel HIP-event timing of every ``every``-th call (events cost ~1.6 % when on every call)."""
        N.check(self._lib.cf_profile_enable(self._handle, int(every) if on else 0))

    def profile_reset(self):
        N.check(self._lib.cf_profile_reset(self._handle))

    def profile_read(self):
        """-> {kernel slot name: (total ms, launches)} since the last reset."""
        ms = (C.c_double * N.CF_PROF_SLOTS)()
        cnt = (C.c_int64 * N.CF_PROF_SLOTS)()
        N.check(self._lib.cf_profile_read(self._handle, ms, cnt))
        return {self._lib.cf_profile_slot_name(i).decode(): (float(ms[i]), int(cnt[i]))
                for i in range(N.CF_PROF_SLOTS) if cnt[i]}

    def debug_stage(self, stage, n_windows):
        feats = self.layer_size_res if stage < self.n_layers_res else 2 * self.layer_size
        out = np.empty((n_windows, WINDOW, feats), dtype=np.float32)
        N.check(self._lib.cf_debug_stage(self._handle, int(stage), int(n_windows), out.ctypes.data_as(C.c_void_p)))
        return out
