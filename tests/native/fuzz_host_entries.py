"""Fuzz driver for the host-only C-ABI entry points under AddressSanitizer + UBSan (run by tests/test_host_sanitizers.py in a
child process with the sanitizer runtime preloaded; not collected by pytest itself).

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tests/native/fuzz_host_entries.py <sanitized .so> <scratch dir>

* cf_load_npy_int16 (catfish_amd/csrc/loader_host.hpp) parses file headers nobody vouches for: well-formed reads, truncated
  ones, wrong versions, huge header lengths, 16-digit shapes, zero-length arrays, other dtypes / orders / ranks, random bytes,
  directories and missing paths -- into exactly-sized buffers.  Judge: numpy's own reader.
* cf_chunks_from_spans / cf_chunks_json (chunks_host.hpp) write into caller-sized buffers: exact, minimum and too-small
  capacities.  Judge: the per-read Python rules (catfish_amd.cli.chunks_of_read) and json.dumps.
* cf_stat_files (loader_host.hpp) writes one size per name from several threads: files, empty files, directories, missing entries,
  zero to 700 names.  Judge: os.stat.
* cf_listing_split_npy_int16 (split_host.hpp) slices reads at coordinates nobody vouches for (below zero, beyond the read, end before
  start) and writes files: random reads, chunk rows and thread counts; names with several dots or none.  Judge: numpy slicing + numpy.save.
* cf_listing_open / _sizes / _names / _close (loader_host.hpp): random name sets against sorted(), blocks into exactly-sized buffers.
A sanitizer report aborts the process (non-zero exit); a wrong answer raises.
"""
import ctypes as C
import io
import json
import os
import sys

import numpy as np
from hypothesis import given, settings, strategies as st, HealthCheck

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = C.CDLL(sys.argv[1])
SCRATCH = sys.argv[2]
P64, P16, P8 = C.POINTER(C.c_int64), C.POINTER(C.c_int16), C.POINTER(C.c_uint8)
lib.cf_last_error.restype = C.c_char_p
lib.cf_load_npy_int16.restype = C.c_int
lib.cf_load_npy_int16.argtypes = [C.c_char_p, P64, C.c_int64, P16, C.c_int64, P64, P64, C.c_int32]
lib.cf_chunks_from_spans.restype = C.c_int
lib.cf_chunks_from_spans.argtypes = [P64, P64, P64, P64, C.c_int64, C.c_int64, P64, P64, P64, C.c_int64, P64, P64, P64, C.c_int64]
lib.cf_chunks_json.restype = C.c_int64
lib.cf_chunks_json.argtypes = [C.c_char_p, P64, C.c_int64, P64, P64, P64, P8, C.c_char_p, C.c_int64]
lib.cf_stat_files.restype = C.c_int
lib.cf_stat_files.argtypes = [C.c_char_p, C.c_char_p, P64, C.c_int64, P64, C.c_int32]
lib.cf_listing_open.restype = C.c_int
lib.cf_listing_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), P64, C.POINTER(C.c_uint64)]
lib.cf_listing_sizes.restype = C.c_int
lib.cf_listing_sizes.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P64, C.c_int32]
lib.cf_listing_names.restype = C.c_int
lib.cf_listing_names.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, P64, P64]
lib.cf_listing_from_names.restype = C.c_int
lib.cf_listing_from_names.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
lib.cf_crc32c.restype = C.c_uint32
lib.cf_crc32c.argtypes = [C.c_char_p, C.c_int64, C.c_uint32]
lib.cf_listing_load_npy_int16.restype = C.c_int
lib.cf_listing_load_npy_int16.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P16, C.c_int64, P64, P64, C.c_int32]
lib.cf_listing_split_npy_int16.restype = C.c_int
lib.cf_listing_split_npy_int16.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P64, P64, P64, P64, P64, P64, C.c_char_p, C.c_char_p, C.c_int32, P64]
lib.cf_listing_close.restype = None
lib.cf_listing_close.argtypes = [C.c_void_p]
CF_OK, CF_ERR_INVALID, CF_ERR_IO = 0, -1, -4
SETTINGS = dict(deadline=None, database=None, suppress_health_check=list(HealthCheck))


def arr64(values):
    """An exactly-sized heap array (the sanitizer's allocator: one element past it is a report)."""
    a = (C.c_int64 * max(len(values), 1))(*values) if len(values) else (C.c_int64 * 1)()
    return a


# ------------------------------------------------------------------------------------------------ cf_load_npy_int16
def npy_bytes(data, version=(1, 0)):
    buf = io.BytesIO()
    np.lib.format.write_array(buf, data, version=version)
    return buf.getvalue()


@st.composite
def file_case(draw):
    kind = draw(st.sampled_from(["good", "good", "good", "empty_array", "truncated", "extended", "version3", "huge_hlen", "digits16",
                                 "two_d", "fortran", "int32", "big_endian", "random", "zero_bytes", "directory", "missing", "magic_only",
                                 "long_header"]))
    n = draw(st.integers(0, 3000))
    data = np.asarray(draw(st.lists(st.integers(-32768, 32767), min_size=min(n, 40), max_size=min(n, 40))) * (n // 40 + 1), np.int16)[:n]
    good = npy_bytes(data, draw(st.sampled_from([(1, 0), (2, 0)])))
    if kind == "good":
        return kind, good
    if kind == "empty_array":
        return kind, npy_bytes(np.zeros(0, np.int16))
    if kind == "truncated":
        return kind, good[:draw(st.integers(0, max(len(good) - 1, 0)))]
    if kind == "extended":
        return kind, good + b"\x00" * draw(st.integers(1, 9))
    if kind == "version3":
        return kind, good[:6] + b"\x03\x00" + good[8:]
    if kind == "huge_hlen":
        v2 = npy_bytes(data, (2, 0))
        return kind, v2[:8] + draw(st.sampled_from([b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\x00\x00\x10\x00"])) + v2[12:]
    if kind == "digits16":
        return kind, good.replace(b"'shape': (%d,)" % len(data), b"'shape': (1234567890123456,)")
    if kind == "two_d":
        return kind, npy_bytes(np.zeros((3, 4), np.int16))
    if kind == "fortran":
        return kind, good.replace(b"'fortran_order': False", b"'fortran_order': True ")
    if kind == "int32":
        return kind, npy_bytes(data.astype(np.int32))
    if kind == "big_endian":
        return kind, npy_bytes(data.astype(">i2"))
    if kind == "random":
        return kind, draw(st.binary(min_size=0, max_size=300))
    if kind == "zero_bytes":
        return kind, b""
    if kind == "magic_only":
        return kind, b"\x93NUMPY\x01\x00" + draw(st.binary(min_size=0, max_size=20))
    if kind == "long_header":             # a VALID file whose header is padded beyond the loader's first 4 KiB read
        head = ("{'descr': '<i2', 'fortran_order': False, 'shape': (%d,), }" % len(data)).ljust(draw(st.integers(4200, 9000)) - 1) + "\n"
        return kind, b"\x93NUMPY\x02\x00" + len(head).to_bytes(4, "little") + head.encode() + data.tobytes()
    return kind, None


def numpy_says(path):
    """What a reader that trusts numpy accepts: a one-dimensional C-order '<i2' array in a file with nothing after it."""
    try:
        if not os.path.isfile(path):
            return None
        with open(path, "rb") as fh:
            version = np.lib.format.read_magic(fh)
            if version not in ((1, 0), (2, 0)):
                return None
            shape, fortran, dtype = np.lib.format._read_array_header(fh, version)
            if fortran or dtype != np.dtype("<i2") or len(shape) != 1:
                return None
            rest = fh.read()
            if shape[0] >= 10 ** 15 or len(rest) != 2 * shape[0]:          # the file ends with the data (infer._read_npy_int16's rule)
                return None
            return np.frombuffer(rest, dtype="<i2")
    except Exception:
        return None


@settings(max_examples=int(os.environ.get("FUZZ_EXAMPLES", 250)), **SETTINGS)
@given(st.lists(file_case(), min_size=1, max_size=6), st.sampled_from(["exact", "short", "roomy"]), st.integers(1, 8))
def fuzz_loader(cases, room, n_threads):
    import shutil
    import tempfile
    box = tempfile.mkdtemp(dir=SCRATCH)
    try:
        _fuzz_loader_in(box, cases, room, n_threads)
    finally:
        shutil.rmtree(box, ignore_errors=True)


def _fuzz_loader_in(box, cases, room, n_threads):
    paths = []
    for i, (kind, blob) in enumerate(cases):
        path = os.path.join(box, "c%d.npy" % i)
        if kind == "directory":
            os.makedirs(path, exist_ok=True)
        elif kind != "missing":
            with open(path, "wb") as fh:
                fh.write(blob)
        paths.append(path)
    want = [numpy_says(p) for p in paths]
    good = all(w is not None for w in want)
    total_want = sum(len(w) for w in want) if good else 0
    cap = {"exact": total_want, "short": max(total_want - 1, 0), "roomy": total_want + 7}[room]
    enc = [os.fsencode(p) for p in paths]
    blob = b"\x00".join(enc) + b"\x00"
    bounds = arr64(list(np.cumsum([0] + [len(e) + 1 for e in enc])))
    out = (C.c_int16 * max(cap, 1))()
    lengths = (C.c_int64 * len(paths))()
    total = C.c_int64(-5)
    rc = lib.cf_load_npy_int16(blob, bounds, len(paths), out if cap else None, cap, lengths, C.byref(total), n_threads)
    if not good:
        assert rc == CF_ERR_INVALID, (rc, [k for k, _ in cases])
        first_bad = next(p for p, w in zip(paths, want) if w is None)
        assert os.fsencode(first_bad) in lib.cf_last_error(), lib.cf_last_error()
    elif cap < total_want:
        assert rc == CF_ERR_INVALID and total.value == total_want
    else:
        assert rc == CF_OK, (rc, lib.cf_last_error(), [k for k, _ in cases])
        assert total.value == total_want and list(lengths) == [len(w) for w in want]
        if total_want:
            assert np.array_equal(np.ctypeslib.as_array(out)[:total_want], np.concatenate(want))


@settings(max_examples=max(20, int(os.environ.get("FUZZ_EXAMPLES", 250)) // 4), **SETTINGS)
@given(st.lists(file_case(), min_size=1, max_size=6), st.sampled_from(["exact", "short", "roomy"]), st.integers(1, 8))
def fuzz_listing_loader(cases, room, n_threads):
    """cf_listing_load_npy_int16 over a whole directory of the loader fuzz's file kinds (named *.npy, read relative to the listing's
    directory): same verdicts and bytes as numpy's reader, exactly-sized buffers."""
    import shutil
    import tempfile
    box = tempfile.mkdtemp(dir=SCRATCH)
    try:
        names = []
        for i, (kind, blob) in enumerate(cases):
            if kind == "missing":
                continue                                  # (a listing cannot hold a missing entry)
            name = "c%d.npy" % i
            if kind == "directory":
                os.makedirs(os.path.join(box, name))
            else:
                with open(os.path.join(box, name), "wb") as fh:
                    fh.write(blob)
            names.append(name)
        names.sort()
        handle, n_entries = C.c_void_p(), C.c_int64(0)
        assert lib.cf_listing_open(os.fsencode(box), C.byref(handle), C.byref(n_entries), None) == CF_OK
        try:
            assert n_entries.value == len(names)
            want = [numpy_says(os.path.join(box, n)) for n in names]
            good = all(w is not None for w in want)
            total_want = sum(len(w) for w in want) if good else 0
            cap = {"exact": total_want, "short": max(total_want - 1, 0), "roomy": total_want + 7}[room]
            out = (C.c_int16 * max(cap, 1))()
            lengths = (C.c_int64 * max(len(names), 1))()
            total = C.c_int64(-5)
            rc = lib.cf_listing_load_npy_int16(handle, 0, len(names), out if cap else None, cap, lengths, C.byref(total), n_threads)
            if not names:
                assert rc == CF_OK and total.value == 0
            elif not good:
                assert rc == CF_ERR_INVALID, (rc, [k for k, _ in cases])
            elif cap < total_want:
                assert rc == CF_ERR_INVALID and total.value == total_want
            else:
                assert rc == CF_OK, (rc, lib.cf_last_error())
                assert total.value == total_want and list(lengths)[:len(names)] == [len(w) for w in want]
                if total_want:
                    assert np.array_equal(np.ctypeslib.as_array(out)[:total_want], np.concatenate(want))
        finally:
            lib.cf_listing_close(handle)
    finally:
        shutil.rmtree(box, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ cf_listing_split_npy_int16
@st.composite
def split_case(draw):
    n_reads = draw(st.integers(1, 7))
    stems = draw(st.lists(st.sampled_from(["r1", "r2", "ch3.read9", "x", "x.y", "", "r10"]), min_size=n_reads, max_size=n_reads, unique=True))
    reads = []
    for stem in stems:
        n = draw(st.integers(0, 400))
        kind = draw(st.sampled_from(["good", "good", "good", "good", "int32"]))
        coord = st.integers(-n - 30, n + 30)
        hp = draw(st.lists(st.tuples(coord, coord), min_size=0, max_size=4))
        non = draw(st.lists(st.tuples(coord, coord), min_size=0, max_size=4))
        reads.append((stem + ".npy", n, kind, hp, non))
    return reads


@settings(max_examples=int(os.environ.get("FUZZ_EXAMPLES", 250)), **SETTINGS)
@given(split_case(), st.integers(1, 8), st.integers(0, 2 ** 31))
def fuzz_split(reads, n_threads, seed):
    """Every read with HP rows cut at its rows, k running on from the HP rows into the non-HP rows; reads without HP rows untouched; a
    read that is not an int16 vector among those to cut -> CF_ERR_INVALID; same-stem reads (x.npy, x.y.npy) overwrite in listing order."""
    import shutil
    import tempfile
    box = tempfile.mkdtemp(dir=SCRATCH)
    try:
        rng = np.random.default_rng(seed)
        for d in ("in", "HP", "nonHP"):
            os.makedirs(os.path.join(box, d))
        reads = sorted(reads, key=lambda r: os.fsencode(r[0]))
        signals = {}
        for name, n, kind, _hp, _non in reads:
            sig = rng.integers(-32768, 32768, size=n).astype(np.int16)
            signals[name] = sig
            with open(os.path.join(box, "in", name), "wb") as fh:
                fh.write(npy_bytes(sig if kind == "good" else sig.astype(np.int32)))
        handle, n_entries = C.c_void_p(), C.c_int64(0)
        assert lib.cf_listing_open(os.fsencode(os.path.join(box, "in")), C.byref(handle), C.byref(n_entries), None) == CF_OK
        try:
            assert n_entries.value == len(reads)
            hb, nb = [0], [0]
            hs, he, ns, ne = [], [], [], []
            for _name, _n, _kind, hp, non in reads:
                hs += [a for a, _b in hp]; he += [b for _a, b in hp]; ns += [a for a, _b in non]; ne += [b for _a, b in non]
                hb.append(len(hs)); nb.append(len(ns))
            counts = arr64([7, 7, 7, 7])
            rc = lib.cf_listing_split_npy_int16(handle, 0, len(reads), arr64(hb), arr64(hs) if hs else None, arr64(he) if he else None,
                                                arr64(nb), arr64(ns) if ns else None, arr64(ne) if ne else None,
                                                os.fsencode(os.path.join(box, "HP")), os.fsencode(os.path.join(box, "nonHP")), n_threads, counts)
            cut = [r for r in reads if r[3]]
            if any(kind != "good" for _name, _n, kind, _hp, _non in cut):
                assert rc == CF_ERR_INVALID and b".npy" in lib.cf_last_error(), (rc, lib.cf_last_error())
                return
            assert rc == CF_OK, (rc, lib.cf_last_error())
            want = {}
            for name, _n, _kind, hp, non in cut:                  # in listing order: a later read of the same stem overwrites
                for k, (a, b) in enumerate(hp + non):
                    want[("HP" if k < len(hp) else "nonHP", "%s_%d.npy" % (name.split(".")[0], k))] = npy_bytes(signals[name][a:b])
            stems = [name.split(".")[0] for name, *_ in cut]
            got = {(d, f): open(os.path.join(box, d, f), "rb").read() for d in ("HP", "nonHP") for f in os.listdir(os.path.join(box, d))}
            assert set(got) == set(want)
            if len(set(stems)) == len(stems):
                assert got == want
                assert list(counts)[:4] == [len(cut), sum(len(r[3]) for r in cut), sum(len(r[4]) for r in cut),
                                            sum(len(signals[r[0]][a:b]) for r in cut for a, b in r[3] + r[4])]
            else:                                               # shared stems: the later read's pieces win where both wrote
                assert all(got[k] == v for k, v in want.items())
        finally:
            lib.cf_listing_close(handle)
    finally:
        shutil.rmtree(box, ignore_errors=True)


# ------------------------------------------------------------------------------------ cf_chunks_from_spans / cf_chunks_json
@st.composite
def reads_with_spans(draw):
    reads = []
    for _ in range(draw(st.integers(0, 6))):
        length = draw(st.integers(1, 5000))
        n = draw(st.integers(0, 7))
        cuts = sorted(draw(st.lists(st.integers(-30, length + 30), min_size=2 * n, max_size=2 * n)))
        reads.append((length, [[cuts[2 * i], cuts[2 * i + 1]] for i in range(n)]))
    return reads


@settings(max_examples=int(os.environ.get("FUZZ_EXAMPLES", 250)), **SETTINGS)
@given(reads_with_spans(), st.integers(1, 1500), st.sampled_from(["product", "exact", "short"]))
def fuzz_chunks(reads, chunk_size, room):
    from catfish_amd.cli import chunks_of_read
    n_reads = len(reads)
    counts = [len(s) for _l, s in reads]
    sb = arr64(list(np.cumsum([0] + counts)))
    flat = [p for _l, s in reads for p in s]
    ss, se = arr64([p[0] for p in flat]), arr64([p[1] for p in flat])
    lens = arr64([l for l, _s in reads])
    want = [chunks_of_read([list(p) for p in s], l, chunk_size) if s else ([], [([(0, l), l])]) for l, s in reads]
    hp_need = sum(len(w[0]) for w in want)
    non_need = sum(len(w[1]) if s else 1 for w, (_l, s) in zip(want, reads))
    n_spans = len(flat)
    # the native merge asks for room for one more complement stretch than a read may end up with: that is its contract
    if room == "product":
        hp_cap, non_cap = n_spans + n_reads, n_spans + 2 * n_reads                  # catfish_amd/chunks.py
    elif room == "exact":
        hp_cap, non_cap = hp_need, sum((len(w[0]) + 1) if s else 1 for w, (_l, s) in zip(want, reads))
    else:
        hp_cap, non_cap = max(hp_need - 1, 0), max(non_need - 1, 0)
    hb, nb = (C.c_int64 * (n_reads + 1))(), (C.c_int64 * (n_reads + 1))()
    hs, he = (C.c_int64 * max(hp_cap, 1))(), (C.c_int64 * max(hp_cap, 1))()
    ns, ne = (C.c_int64 * max(non_cap, 1))(), (C.c_int64 * max(non_cap, 1))()
    rc = lib.cf_chunks_from_spans(sb, ss, se, lens, n_reads, chunk_size, hb, hs if hp_cap else None, he if hp_cap else None, hp_cap,
                                  nb, ns if non_cap else None, ne if non_cap else None, non_cap)
    if room == "short" and (hp_need > hp_cap or non_need > non_cap):
        assert rc == CF_ERR_INVALID, (rc, hp_need, hp_cap, non_need, non_cap)
        return
    if rc != CF_OK:                     # "short" that happened to fit everything but the spare complement slot
        assert room == "short" and rc == CF_ERR_INVALID
        return
    hp_dict, non_dict, names = {}, {}, []
    for r, (w, (l, s)) in enumerate(zip(want, reads)):
        name = "read %d \"q\" \\ é" % r if r % 2 else "read_%d.fast5" % r
        names.append(name)
        got_hp = [[hs[i], he[i]] for i in range(hb[r], hb[r + 1])]
        got_non = [[ns[i], ne[i]] for i in range(nb[r], nb[r + 1])]
        assert got_hp == [list(p) for p in w[0]], (r, got_hp, w[0])
        if s:
            assert got_non == [list(p) for p in w[1]], (r, got_non, w[1])
            hp_dict[name], non_dict[name] = w[0], w[1]
        else:
            assert got_non == [[0, l]]
            non_dict[name] = [([(0, l), l])]
    # the JSON text of both tables, into the capacity the product asks for, then into every smaller round number
    enc = [json.dumps(n).encode("ascii") for n in names]
    keys = b"".join(enc) or b"\x00"
    kb = arr64(list(np.cumsum([0] + [len(e) for e in enc])))
    whole = (C.c_uint8 * max(n_reads, 1))(*[0 if s else 1 for _l, s in reads])
    for bounds, start, end, flag, ref in ((hb, hs, he, None, hp_dict), (nb, ns, ne, whole, non_dict)):
        rows = bounds[n_reads]
        cap = sum(len(e) for e in enc) + 48 * rows + 40 * n_reads + 64
        text = json.dumps(ref)[1:-1].encode("ascii")
        for c in (cap, len(text) + 96, len(text), len(text) // 2, 0):
            buf = C.create_string_buffer(max(c, 1))
            n = lib.cf_chunks_json(keys, kb, n_reads, bounds, start, end, flag, buf, c)
            assert n == len(text) or n == CF_ERR_INVALID, (n, len(text), c)
            if c == cap:
                assert n == len(text) and buf.raw[:n] == text, (buf.raw[:n], text)
            elif n >= 0:
                assert buf.raw[:n] == text


# ------------------------------------------------------------------------------------------------ cf_stat_files
@settings(max_examples=max(20, int(os.environ.get("FUZZ_EXAMPLES", 250)) // 5), **SETTINGS)
@given(st.lists(st.tuples(st.sampled_from(["file", "file", "file", "empty", "dir", "missing"]), st.integers(0, 5000)), min_size=0, max_size=700),
       st.integers(-1, 9))
def fuzz_stat(entries, n_threads):
    """Sizes of directory entries into an exactly-sized table, from 1..9 threads (blocks of 256 names per thread at most); a missing
    entry is an error that names it.  Judge: os.stat."""
    import shutil
    import tempfile
    box = tempfile.mkdtemp(dir=SCRATCH)
    try:
        names = []
        for i, (kind, size) in enumerate(entries):
            name = "e%04d %s.npy" % (i, "\u00e9" if i % 3 == 0 else "x")
            path = os.path.join(box, name)
            if kind == "file":
                with open(path, "wb") as fh:
                    fh.write(b"\x00" * size)
            elif kind == "empty":
                open(path, "wb").close()
            elif kind == "dir":
                os.makedirs(path)
            names.append(name)
        enc = [os.fsencode(n) for n in names]
        blob = b"\x00".join(enc) + b"\x00"
        bounds = arr64(list(np.cumsum([0] + [len(e) + 1 for e in enc])))
        sizes = (C.c_int64 * max(len(names), 1))()
        rc = lib.cf_stat_files(os.fsencode(box), blob, bounds, len(names), sizes, n_threads)
        missing = [n for n, (kind, _s) in zip(names, entries) if kind == "missing"]
        if missing:
            assert rc == CF_ERR_INVALID and any(os.fsencode(m) in lib.cf_last_error() for m in missing), lib.cf_last_error()
        else:
            assert rc == CF_OK, lib.cf_last_error()
            assert list(sizes)[:len(names)] == [os.stat(os.path.join(box, n)).st_size for n in names]
    finally:
        shutil.rmtree(box, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ cf_listing_*
_NAME = st.builds(lambda p, t: p + t, st.sampled_from(["", "r", "read_", "read_00000000", "ch_"]),
                  st.text(st.sampled_from(list("0123456789abXY_.~") + ["\u00e9", "\u4e2d"]), min_size=1, max_size=20)).filter(
                      lambda n: n not in (".", ".."))


@settings(max_examples=max(20, int(os.environ.get("FUZZ_EXAMPLES", 250)) // 4), **SETTINGS)
@given(st.sets(_NAME, min_size=0, max_size=80), st.data())
def fuzz_listing(names, data):
    """Names in, order out: against sorted(); every block's names into an exactly-sized buffer (and a too-small one), every block's
    sizes into an exactly-sized table, ranges off either end refused."""
    import shutil
    import tempfile
    box = tempfile.mkdtemp(dir=SCRATCH)
    try:
        for i, n in enumerate(sorted(names)):
            with open(os.path.join(box, n), "wb") as fh:
                fh.write(b"\x01" * (i % 9))
        want = sorted(names)
        handle, n_entries, digest = C.c_void_p(), C.c_int64(-1), (C.c_uint64 * 2)()
        assert lib.cf_listing_open(os.fsencode(box), C.byref(handle), C.byref(n_entries), digest) == CF_OK, lib.cf_last_error()
        try:
            assert n_entries.value == len(want)
            lo = data.draw(st.integers(0, len(want)))
            hi = data.draw(st.integers(lo, len(want)))
            need = C.c_int64(-1)
            assert lib.cf_listing_names(handle, lo, hi, None, 0, None, C.byref(need)) == CF_OK
            enc = [os.fsencode(n) for n in want[lo:hi]]
            assert need.value == sum(len(e) + 1 for e in enc)
            buf = C.create_string_buffer(max(need.value, 1))
            bounds = (C.c_int64 * (hi - lo + 1))()
            assert lib.cf_listing_names(handle, lo, hi, buf, need.value, bounds, None) == CF_OK
            assert buf.raw[:need.value] == b"".join(e + b"\x00" for e in enc)
            assert list(bounds) == list(np.cumsum([0] + [len(e) + 1 for e in enc]))
            if need.value:
                small = C.create_string_buffer(max(need.value - 1, 1))
                assert lib.cf_listing_names(handle, lo, hi, small, need.value - 1, bounds, None) == CF_ERR_INVALID
            sizes = (C.c_int64 * max(hi - lo, 1))()
            assert lib.cf_listing_sizes(handle, lo, hi, sizes, data.draw(st.integers(-1, 6))) == CF_OK, lib.cf_last_error()
            assert list(sizes)[:hi - lo] == [os.stat(os.path.join(box, n)).st_size for n in want[lo:hi]]
            assert lib.cf_listing_sizes(handle, -1, hi, sizes, 1) == CF_ERR_INVALID
            assert lib.cf_listing_sizes(handle, lo, len(want) + 1, sizes, 1) == CF_ERR_INVALID
            assert lib.cf_listing_names(handle, hi + 1, hi, None, 0, None, None) == CF_ERR_INVALID
            # the ordered names as they travel between ranks -> the same listing; damaged blobs are refused (exactly-sized copies)
            assert lib.cf_listing_names(handle, 0, len(want), None, 0, None, C.byref(need)) == CF_OK
            whole = C.create_string_buffer(max(need.value, 1))
            all_bounds = (C.c_int64 * (len(want) + 1))()
            assert lib.cf_listing_names(handle, 0, len(want), whole, need.value, all_bounds, None) == CF_OK
            blob = whole.raw[:need.value]
            twin, d2 = C.c_void_p(), (C.c_uint64 * 2)()
            assert lib.cf_listing_from_names(os.fsencode(box), blob, len(blob), len(want), C.byref(twin), d2) == CF_OK, lib.cf_last_error()
            assert list(d2) == list(digest)
            lib.cf_listing_close(twin)
            if want:
                for damaged, n in ((blob[:-1], len(want)), (blob, len(want) + 1), (blob, len(want) - 1), (blob + blob, 2 * len(want)),
                                   (b"\x00" + blob, len(want) + 1)):
                    t2 = C.c_void_p()
                    assert lib.cf_listing_from_names(os.fsencode(box), bytes(damaged), len(damaged), n, C.byref(t2), None) == CF_ERR_INVALID
                    assert not t2.value
        finally:
            lib.cf_listing_close(handle)
    finally:
        shutil.rmtree(box, ignore_errors=True)


@settings(max_examples=max(20, int(os.environ.get("FUZZ_EXAMPLES", 250)) // 2), **SETTINGS)
@given(st.binary(min_size=0, max_size=600), st.integers(0, 2 ** 32 - 1), st.integers(0, 9))
def fuzz_crc(data, seed, skip):
    """cf_crc32c on exactly-sized buffers at every alignment (reads past the end are a sanitizer report) against the byte loop."""
    from catfish_amd.checkpoint import crc32c_python
    piece = data[min(skip, len(data)):]
    assert lib.cf_crc32c(piece, len(piece), seed) == crc32c_python(piece, seed)


def bad_arguments():
    """NULL tables, negative sizes, descending bounds: an error code, never a fault."""
    z = arr64([0, 0])
    one = arr64([1])
    assert lib.cf_load_npy_int16(None, None, -1, None, 0, None, None, 1) == CF_ERR_INVALID
    assert lib.cf_load_npy_int16(None, None, 0, None, 0, None, None, 1) == CF_OK
    assert lib.cf_load_npy_int16(b"x\x00", None, 1, None, 0, None, None, 1) == CF_ERR_INVALID
    assert lib.cf_chunks_from_spans(None, None, None, None, 0, 10, None, None, None, 0, None, None, None, 0) == CF_ERR_INVALID
    assert lib.cf_chunks_from_spans(z, None, None, one, 1, 10, z, None, None, 0, z, None, None, 1) == CF_ERR_INVALID      # null output
    assert lib.cf_chunks_from_spans(arr64([2, 0]), one, one, one, 1, 10, z, one, one, 1, z, one, one, 1) == CF_ERR_INVALID  # descending
    assert lib.cf_chunks_from_spans(z, None, None, one, -3, 10, z, None, None, 0, z, None, None, 0) == CF_ERR_INVALID
    buf = C.create_string_buffer(8)
    assert lib.cf_chunks_json(b'"a"', arr64([3, 0]), 1, arr64([0, 1]), one, one, None, buf, 8) == CF_ERR_INVALID            # negative key length
    assert lib.cf_chunks_json(b'"a"', arr64([0, 3]), 1, arr64([0, 1]), None, None, None, buf, 8) == CF_ERR_INVALID          # rows without tables
    assert lib.cf_chunks_json(None, z, 0, z, None, None, None, buf, -1) == CF_ERR_INVALID
    assert lib.cf_stat_files(None, None, None, 0, None, 1) == CF_OK
    assert lib.cf_stat_files(None, None, None, 2, None, 1) == CF_ERR_INVALID
    assert lib.cf_stat_files(b".", b"x\x00", z, -1, z, 1) == CF_ERR_INVALID
    assert lib.cf_stat_files(os.fsencode(os.path.join(SCRATCH, "no such directory")), b"x\x00", z, 1, arr64([0]), 1) == CF_ERR_INVALID
    assert b"no such directory" in lib.cf_last_error()
    h = C.c_void_p()
    assert lib.cf_listing_open(None, C.byref(h), None, None) == CF_ERR_INVALID
    assert lib.cf_listing_open(os.fsencode(os.path.join(SCRATCH, "no such directory")), C.byref(h), None, None) == CF_ERR_INVALID and not h.value
    assert lib.cf_listing_sizes(None, 0, 0, None, 1) == CF_ERR_INVALID and lib.cf_listing_names(None, 0, 0, None, 0, None, None) == CF_ERR_INVALID
    lib.cf_listing_close(None)
    assert lib.cf_listing_split_npy_int16(None, 0, 0, None, None, None, None, None, None, None, None, 1, None) == CF_ERR_INVALID
    assert lib.cf_listing_open(os.fsencode(SCRATCH), C.byref(h), None, None) == CF_OK
    assert lib.cf_listing_split_npy_int16(h, 0, 0, None, None, None, None, None, None, None, None, 1, None) == CF_OK          # nothing to do
    lib.cf_listing_close(h)
    assert lib.cf_crc32c(None, 0, 5) == 5 and lib.cf_crc32c(None, 10, 7) == 7 and lib.cf_crc32c(b"123456789", 9, 0) == 0xE3069283


if __name__ == "__main__":
    bad_arguments()
    fuzz_loader()
    fuzz_chunks()
    fuzz_stat()
    fuzz_listing()
    fuzz_listing_loader()
    fuzz_crc()
    fuzz_split()
    print("fuzz ok")
