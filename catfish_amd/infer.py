"""Read-level inference -- the reference's catfish/infer.py surface on the MI355X engine.

Function names, arguments, return types and error behaviour follow the reference
(catfish/infer.py:12-198); the three per-sample Python loops of the reference's
post-processing (:138, :153-157, :186-190; ~2 M samples/s/core) are replaced by
vectorised run-length arithmetic with identical results (tests/golden/postproc_golden.json
holds outputs of the reference's own functions).
"""
from __future__ import annotations

import os

import numpy as np

WINDOW_SIZE = 35


# --------------------------------------------------------------------------- raw signal
def normalize_raw_signal(raw, norm_method):
    """infer.py:96-105: (raw - median) / median(|raw - median|); unknown method -> ValueError."""
    if norm_method != "median":
        raise ValueError("norm_method not recognized")
    raw = np.asarray(raw)
    shift = np.median(raw)
    scale = np.median(np.abs(raw - shift))
    return (raw - shift) / scale


def _trimmed_fast5_signal(fast5_file):
    """infer.py:87-90: the raw DAC samples of the single read after the leader (``first_sample_template``)."""
    first_sample = fast5_file["Analyses/Segmentation_000/Summary/segmentation"].attrs["first_sample_template"]
    read_name = fast5_file["Raw/Reads/"].visit(str)
    raw_signal = fast5_file["Raw/Reads/" + read_name + "/Signal"][()]
    return raw_signal[first_sample:]


def process_signal(fast5_file, normalization="median"):
    """infer.py:77-93: trim the leader (``first_sample_template``) and normalise.

    ``fast5_file`` is an open h5py.File, exactly as in the reference.
    """
    return normalize_raw_signal(_trimmed_fast5_signal(fast5_file), normalization)


def _read_npy_int16(path):
    """Fast path for the common case of a directory of reads: a version-1/2 ``.npy`` holding a one-dimensional C-order
    little-endian int16 array, read with one ``read()`` and no literal_eval (np.load spends ~80 us per file in its
    header parser, which bounds a rank at ~50 M samples/s).  Anything else returns None and goes through np.load."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 12 or buf[:6] != b"\x93NUMPY" or buf[6] not in (1, 2):
        return None
    if buf[6] == 1:
        hlen, off = int.from_bytes(buf[8:10], "little"), 10
    else:
        hlen, off = int.from_bytes(buf[8:12], "little"), 12
    header = buf[off:off + hlen]
    if b"'descr': '<i2'" not in header or b"'fortran_order': False" not in header:
        return None
    a, b = header.find(b"'shape': ("), header.find(b")", header.find(b"'shape': ("))
    if a < 0 or b < 0:
        return None
    dims = [d for d in header[a + 10:b].split(b",") if d.strip()]
    if len(dims) != 1 or not dims[0].strip().isdigit():
        return None
    n = int(dims[0])
    if len(buf) != off + hlen + 2 * n:
        return None
    return np.frombuffer(buf, dtype="<i2", count=n, offset=off + hlen)


def load_dac(path):
    """Raw samples of one read after the leader trim, NOT normalised (what the device ingest path uploads).

    ``.fast5`` needs h5py (reference behaviour, infer.py:27-29); because h5py/libhdf5 are not
    part of the MI355X image, ``.npy`` (int16 DAC after the leader trim), ``.npz`` (key ``raw``
    or ``signal``; the reference's NPZ layout, networks/reader.py:11-23) and headerless
    little-endian int16 ``.bin``/``.raw`` files are accepted too.
    """
    if not os.path.exists(path):
        raise ValueError("path to FAST5 is not correct.")      # infer.py:25-26
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        raw = _read_npy_int16(path)
        if raw is None:
            raw = np.load(path, allow_pickle=False)
    elif ext == ".npz":
        with np.load(path, allow_pickle=False) as z:
            raw = z["raw"] if "raw" in z.files else z["signal"]
    elif ext in (".bin", ".raw"):
        raw = np.fromfile(path, dtype="<i2")
    else:
        try:
            import h5py
        except ImportError:
            raise ImportError("reading %s needs h5py, which is not installed; convert the read to "
                              ".npy/.npz/.bin (int16 DAC samples)" % path)
        with h5py.File(path, "r") as fast5:
            raw = _trimmed_fast5_signal(fast5)
    return np.asarray(raw).reshape(-1)


def is_dac(raw):
    """True when ``raw`` can go up as int16 DAC codes unchanged (the device normalisation is exact on those)."""
    raw = np.asarray(raw)
    if raw.dtype == np.int16:
        return True
    if raw.dtype.kind in "iu" and raw.size:
        return int(raw.min()) >= -32768 and int(raw.max()) <= 32767
    return raw.dtype.kind in "iu"


def load_raw(path):
    """One read, trimmed and normalised (process_signal's result for any of ``load_dac``'s formats)."""
    return normalize_raw_signal(load_dac(path), "median")


def padding_size_for(length, window_size=WINDOW_SIZE):
    """infer.py:32-36: pad to a multiple of the window; a multiple gets a FULL extra window."""
    if not (length / window_size).is_integer():
        return window_size - (length - (length // window_size * window_size))
    return 35


def reshape_input(data, window, n_inputs):
    """infer.py:108-124 (a failed reshape is printed and swallowed there; same here)."""
    try:
        data = np.reshape(data, (-1, window, n_inputs))
    except ValueError:
        print(len(data))
        print(len(data[0]))
    return data


# --------------------------------------------------------------------------- classified output
def class_from_threshold(predicted_scores, threshold=0.5):
    """infer.py:128-138 -> list of ints."""
    return (np.asarray(predicted_scores) >= threshold).astype(np.int64).tolist()


def _run_length(values):
    """values[n] -> (run values, run starts, run lengths)."""
    v = np.asarray(values)
    n = v.shape[0]
    if n == 0:
        raise IndexError("list index out of range")   # the reference indexes predictions[0]
    starts = np.concatenate(([0], np.flatnonzero(v[1:] != v[:-1]) + 1))
    lengths = np.diff(np.concatenate((starts, [n])))
    return v[starts], starts, lengths


def correct_short(predictions, threshold=15):
    """infer.py:174-198: non-zero runs shorter than ``threshold`` are set to 0."""
    vals, _starts, lengths = _run_length(predictions)
    vals = np.where((vals != 0) & (lengths < threshold), 0, vals)
    return np.repeat(vals, lengths)


def hp_in_pred(predictions, extension_left=11, extension_right=16, label=1):
    """infer.py:141-162: each run of ``label`` -> [start - 11, start + length + 16] (end exclusive;
    spans may start below 0 or end beyond the read, exactly like the reference)."""
    vals, starts, lengths = _run_length(predictions)
    sel = vals == label
    return [[int(s) - extension_left, int(s) + int(ln) + extension_right]
            for s, ln in zip(starts[sel], lengths[sel])]


# --------------------------------------------------------------------------- inference
def infer_class_from_raw(raw, model, label=1, window_size=WINDOW_SIZE):
    """Body of infer_class_from_signal (infer.py:31-51) for an already normalised signal."""
    raw = np.asarray(raw)
    padding_size = padding_size_for(len(raw), window_size)
    raw = np.hstack((raw, np.array(padding_size * [0])))
    raw_in = reshape_input(raw, window_size, 1)
    scores = model.infer(raw_in)
    scores = scores[:-padding_size]
    labels = correct_short(class_from_threshold(scores))
    predicted_hps = hp_in_pred(labels)
    return predicted_hps, len(labels)


def infer_class_from_signal(fast5_file, model, label=1, window_size=WINDOW_SIZE):
    """infer.py:12-51.  Returns (list of [start, end] homopolymer spans, length of the read)."""
    raw = load_raw(fast5_file)
    return infer_class_from_raw(raw, model, label=label, window_size=window_size)


def infer_class_from_npz(npz_file, model, label=1, window_size=WINDOW_SIZE):
    """infer.py:54-73: spans of the TRUE labels stored in an NPZ (no network involved)."""
    if not os.path.exists(npz_file):
        raise ValueError("path to NPZ is not correct.")
    with np.load(npz_file, allow_pickle=False) as z:
        labels = z["base_labels"]
    return hp_in_pred(labels, 0, 0), len(labels)
