"""The split step of the ``catfish`` pipeline (catfish/catfish:85-92 -> catfish/split_f5.py:8-81) for the read formats this
image can hold.

The reference cuts the raw signal of every read THAT HAS homopolymer chunks (``for read in hp_dict``, catfish/catfish:88) at
the chunk coordinates: ``Signal[s[0]:s[1]]`` of each merged HP chunk goes to ``TEMP/HP/<stem>_<k>.fast5``, of each non-HP
stretch to ``TEMP/nonHP/<stem>_<k>.fast5``; ``<stem>`` is the file name up to its FIRST dot (split_f5.py:39,65) and ``k``
counts on from the HP chunks into the non-HP ones (split_f5.py:34,57,81: one ``index`` for both loops).  Its container is a
gzip-9 HDF5 copy of the input with the signal replaced; HDF5 is outside this path, so here the slice
is written as the same int16 samples in the format the reads came in: a one-dimensional little-endian int16 ``.npy``
(byte for byte ``numpy.save`` of ``signal[s0:s1].astype('<i2')``), which ``infer.load_dac`` reads back.  Slices follow
Python / numpy rules (a start below 0 counts from the end, an end beyond the read stops at it), which is what h5py applies to
``signal_dset[s[0]:s[1]]``.  ``.fast5`` input keeps raising ``infer.load_dac``'s ImportError.

Every rank of a sharded job splits the reads it classified (``split_reads`` over its own ``ChunkTable`` rows): no gather.  A block
of int16 ``.npy`` reads goes through the C-ABI library's host thread pool (``cf_listing_split_npy_int16``, csrc/split_host.hpp: same
files, byte for byte); the Python loop here is the definition and serves every other format.
"""
from __future__ import annotations

import os

import numpy as np

from .infer import load_dac

_HEADERS = {}


def npy_header(n):
    """The 128-byte (64-aligned) version-1.0 header ``numpy.save`` writes for an int16 vector of ``n`` samples."""
    h = _HEADERS.get(n)
    if h is None:
        text = "{'descr': '<i2', 'fortran_order': False, 'shape': (%d,), }" % n
        pad = -(10 + len(text) + 1) % 64
        h = b"\x93NUMPY\x01\x00" + (len(text) + pad + 1).to_bytes(2, "little") + text.encode("latin1") + b" " * pad + b"\n"
        if len(_HEADERS) < 4096:
            _HEADERS[n] = h
    return h


def chunk_stem(input_file):
    """split_f5.py:39: ``os.path.basename(input_file).split(".")[0]`` -- up to the FIRST dot, so ``a.b.npy`` and ``a.c.npy``
    share their chunk names (and the later read overwrites the earlier one's files, as in the reference)."""
    return os.path.basename(input_file).split(".")[0]


def _write_chunk(dest_name, signal, s0, s1):
    part = np.ascontiguousarray(signal[int(s0):int(s1)], dtype="<i2")       # split_f5.py:36,45: dtype="int16"
    with open(dest_name, "wb") as fh:
        fh.write(npy_header(part.shape[0]) + part.tobytes())
    return part.shape[0]


def split_signal(input_file, splits_hp, splits_nonhp, temp_dir, temp_dir_nonhp, signal=None):
    """split_f5.split_signal (:8-81) for one read: -> (HP chunk files, non-HP chunk files) in writing order (what the
    reference's docstring promises; its code returns the HDF5 read name once per file).  ``signal``: the read's samples
    when the caller holds them already, else ``infer.load_dac(input_file)`` (ValueError for a wrong path, ImportError for a
    ``.fast5`` without h5py -- the reference's IOError / RuntimeError cases)."""
    if signal is None:
        signal = load_dac(input_file)
    signal = np.asarray(signal).reshape(-1)
    stem = chunk_stem(input_file)
    hp_list, nonhp_list = [], []
    index = 0
    for out, directory, splits in ((hp_list, temp_dir, splits_hp), (nonhp_list, temp_dir_nonhp, splits_nonhp)):
        for s in splits:
            dest_name = "{}/{}_{}.npy".format(directory, stem, index)
            _write_chunk(dest_name, signal, s[0], s[1])
            index += 1
            out.append(dest_name)
    return hp_list, nonhp_list


def split_listing(table, listing, lo, temp_dir_hp, temp_dir_nonhp, n_threads=None):
    """``split_reads`` for entries [lo, lo + len(table)) of a ``sharding.DirListing`` through the library's host thread pool
    (``cf_listing_split_npy_int16``: every input read once, every piece one ``write``; no Python per file -- which costs ~30 us a
    piece, ten times the classification of the same reads).  int16 ``.npy`` reads only: ValueError names the first entry of another
    kind (the caller repeats the block through ``split_reads``; pieces are rewritten whole), OSError the first piece that could
    not be written."""
    import ctypes as C
    from . import _native as N
    counts = np.zeros(4, np.int64)

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)
    if n_threads is None:
        n_threads = 4          # creating files serialises on the output directory's lock: nothing is gained past 2-4 (tmpfs: 139 k / 198 k / 192 k files/s with 1 / 2 / 4)
    N.check(N.lib().cf_listing_split_npy_int16(listing._handle, int(lo), int(lo) + len(table), ptr(table.hp_bounds), ptr(table.hp_start),
                                               ptr(table.hp_end), ptr(table.nonhp_bounds), ptr(table.nonhp_start), ptr(table.nonhp_end),
                                               os.fsencode(temp_dir_hp), os.fsencode(temp_dir_nonhp), int(n_threads), ptr(counts)))
    return dict(zip(("reads", "files_hp", "files_nonhp", "samples"), (int(c) for c in counts)))


def split_reads(table, paths, temp_dir_hp, temp_dir_nonhp, listing=None, lo=0):
    """catfish/catfish:86-89 over a ``chunks.ChunkTable``: every read with homopolymer rows is cut at its HP rows, then at its
    non-HP rows (reads without any are not in the reference's ``hp_dict`` and are not split).  ``paths``: one per table row.
    With ``listing`` / ``lo`` (the rows are entries [lo, lo + len) of that ``DirListing``) the native pool does it when every such
    read is an int16 ``.npy``; otherwise, and for every other format, the loop below.
    -> dict(reads, files_hp, files_nonhp, samples): counts of what was written."""
    if len(paths) != len(table):
        raise ValueError("one path per read")
    if listing is not None and len(table):
        try:
            return split_listing(table, listing, lo, temp_dir_hp, temp_dir_nonhp)
        except ValueError:                                # an entry of another kind (.npz, .bin, int32 codes ...): the general loop
            pass
    hb, nb = table.hp_bounds.tolist(), table.nonhp_bounds.tolist()
    hp_rows = np.stack([table.hp_start, table.hp_end], axis=1).tolist()
    non_rows = np.stack([table.nonhp_start, table.nonhp_end], axis=1).tolist()
    done = {"reads": 0, "files_hp": 0, "files_nonhp": 0, "samples": 0}
    for r in np.flatnonzero(table.has_hp).tolist():
        path = paths[r]
        signal = np.asarray(load_dac(path)).reshape(-1)
        stem, index = chunk_stem(path), 0
        for key, directory, rows in (("files_hp", temp_dir_hp, hp_rows[hb[r]:hb[r + 1]]),
                                     ("files_nonhp", temp_dir_nonhp, non_rows[nb[r]:nb[r + 1]])):
            for s0, s1 in rows:
                done["samples"] += _write_chunk("{}/{}_{}.npy".format(directory, stem, index), signal, s0, s1)
                index += 1
            done[key] += len(rows)
        done["reads"] += 1
    return done
