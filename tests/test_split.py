"""The split step (catfish/catfish:85-92 -> catfish/split_f5.py:8-81) for int16 ``.npy`` / ``.npz`` / ``.bin`` reads.

Golden vectors: tests/golden/split_golden.{json,npz}, made by tests/golden/make_split_golden.py, which EXECUTES the
reference's ``split_signal`` against a stand-in h5py and records, per case, the files it created (directory, name, order)
and the samples of each new Signal dataset.  The product writes ``.npy`` where the reference writes ``.fast5`` (HDF5 is
outside this path): names are compared with the extension swapped, contents sample for sample and -- as files -- byte for
byte with ``numpy.save`` of the same slice."""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from catfish_amd import chunks, infer, split  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "split_golden.json")) as fh:
        doc = json.load(fh)
    arrays = np.load(os.path.join(GOLDEN, "split_golden.npz"))
    return doc, arrays


def _npy(name):
    return name[:-len(".fast5")] + ".npy" if name.endswith(".fast5") else name


def _saved(a):
    buf = io.BytesIO()
    np.save(buf, np.ascontiguousarray(a, dtype="<i2"))
    return buf.getvalue()


def test_split_signal_matches_the_reference_executed_goldens(golden, tmp_path):
    doc, arrays = golden
    assert len(doc) >= 15
    for i, case in enumerate(doc):
        root = tmp_path / ("case%d" % i)
        (root / "in").mkdir(parents=True)
        (root / "HP").mkdir()
        (root / "nonHP").mkdir()
        signal = arrays["signal_%d" % i]
        src = str(root / "in" / _npy(case["name"]))
        if src.endswith(".npy"):
            np.save(src, signal)
            hp_files, non_files = split.split_signal(src, case["hp"], case["nonhp"], str(root / "HP"), str(root / "nonHP"))
        else:                                                      # a name without extension: the caller hands the samples over
            hp_files, non_files = split.split_signal(src, case["hp"], case["nonhp"], str(root / "HP"), str(root / "nonHP"),
                                                     signal=signal)
        want = [(folder, _npy(name)) for folder, name in case["files"]]
        got = [(os.path.basename(os.path.dirname(p)), os.path.basename(p)) for p in hp_files + non_files]
        assert got == want                                         # names, directories and the running index, in writing order
        assert all(f == "HP" for f, _n in got[:len(hp_files)]) and all(f == "nonHP" for f, _n in got[len(hp_files):])
        for j, (folder, name) in enumerate(want):
            data = (root / folder / name).read_bytes()
            expected = arrays["out_%d_%d" % (i, j)]
            assert data == _saved(expected)                        # byte for byte numpy.save of the reference's new Signal
            back = infer.load_dac(str(root / folder / name))       # and a read the pipeline takes back in
            assert back.dtype == np.int16 and np.array_equal(back, expected)
        assert sorted(os.listdir(root / "HP")) == sorted(n for f, n in want if f == "HP")
        assert sorted(os.listdir(root / "nonHP")) == sorted(n for f, n in want if f == "nonHP")


def test_slices_follow_numpy_rules_on_the_edge_cases(golden):
    doc, arrays = golden
    seen = set()
    for i, case in enumerate(doc):
        signal = arrays["signal_%d" % i]
        for j, (s0, s1) in enumerate(case["hp"] + case["nonhp"]):
            assert np.array_equal(arrays["out_%d_%d" % (i, j)], signal[s0:s1])
            seen.add(("neg" if s0 < 0 else "") + ("over" if s1 > len(signal) else "") + ("empty" if len(signal[s0:s1]) == 0 else ""))
    assert {"neg", "over", "empty"} <= {k for key in seen for k in ("neg", "over", "empty") if k in key}


def test_split_reads_over_a_chunk_table_equals_the_per_read_function(golden, tmp_path):
    """The bulk form the CLI uses: reads without homopolymer rows are skipped (``for read in hp_dict``, catfish/catfish:88)."""
    with open(os.path.join(GOLDEN, "postproc_golden.json")) as fh:
        merges = json.load(fh)["merge"]
    rng = np.random.default_rng(5)
    reads, spans, lengths = tmp_path / "reads", [], []
    reads.mkdir()
    paths = []
    for i, c in enumerate(merges[:12]):
        sig = rng.integers(0, 2048, size=c["len_read"]).astype(np.int16)
        ext = (".npy", ".npz", ".bin")[i % 3]
        p = str(reads / ("r%02d%s" % (i, ext)))
        if ext == ".npy":
            np.save(p, sig)
        elif ext == ".npz":
            np.savez(p, raw=sig)
        else:
            sig.tofile(p)
        paths.append(p)
        spans.append([] if i % 4 == 3 else c["spans"])             # every fourth read has no homopolymer at all
        lengths.append(c["len_read"])
    bounds = np.cumsum([0] + [len(s) for s in spans])
    flat = np.array([r for s in spans for r in s], dtype=np.int64).reshape(-1, 2)
    table = chunks.ChunkTable.from_spans(bounds, flat[:, 0], flat[:, 1], lengths, 1000)
    for d in ("a/HP", "a/nonHP", "b/HP", "b/nonHP"):
        (tmp_path / d).mkdir(parents=True)
    done = split.split_reads(table, paths, str(tmp_path / "a/HP"), str(tmp_path / "a/nonHP"))
    hp_dict, nonhp_dict = table.to_dicts(paths)
    for p in hp_dict:                                              # the reference's loop, one read at a time
        split.split_signal(p, hp_dict[p], nonhp_dict[p], str(tmp_path / "b/HP"), str(tmp_path / "b/nonHP"))
    assert done["reads"] == len(hp_dict) == 9
    for d in ("HP", "nonHP"):
        names = sorted(os.listdir(tmp_path / "a" / d))
        assert names == sorted(os.listdir(tmp_path / "b" / d)) and names
        assert all((tmp_path / "a" / d / n).read_bytes() == (tmp_path / "b" / d / n).read_bytes() for n in names)
    assert done["files_hp"] == len(os.listdir(tmp_path / "a/HP")) and done["files_nonhp"] == len(os.listdir(tmp_path / "a/nonHP"))
    assert done["samples"] == sum(len(np.load(tmp_path / "a" / d / n)) for d in ("HP", "nonHP") for n in os.listdir(tmp_path / "a" / d))
    with pytest.raises(ValueError):
        split.split_reads(table, paths[:-1], str(tmp_path / "a/HP"), str(tmp_path / "a/nonHP"))


def test_fast5_input_and_wrong_paths_raise_like_the_loader(tmp_path):
    fast5 = tmp_path / "read.fast5"
    fast5.write_bytes(b"\x89HDF\r\n\x1a\n")
    with pytest.raises(ImportError):                               # no h5py in this image: the existing loader error, unchanged
        split.split_signal(str(fast5), [[0, 5]], [], str(tmp_path), str(tmp_path))
    with pytest.raises(ValueError):
        split.split_signal(str(tmp_path / "absent.npy"), [[0, 5]], [], str(tmp_path), str(tmp_path))
    np.save(tmp_path / "ok.npy", np.arange(10, dtype=np.int16))
    with pytest.raises(OSError):                                   # the output directory is missing: an exception aborts, as in the reference
        split.split_signal(str(tmp_path / "ok.npy"), [[0, 5]], [], str(tmp_path / "nowhere"), str(tmp_path))


def test_npy_header_is_numpys_for_every_length_class():
    for n in (0, 1, 9, 10, 99, 100, 999999, 10**7, 10**12):
        a = np.lib.format.header_data_from_array_1_0(np.zeros(0, "<i2"))
        a["shape"] = (n,)
        buf = io.BytesIO()
        np.lib.format.write_array_header_1_0(buf, a)
        assert split.npy_header(n) == buf.getvalue()


def _table_from_cases(cases):
    hb, nb, hs, he, ns, ne, lens = [0], [0], [], [], [], [], []
    for c in cases:
        hs += [a for a, _b in c["hp"]]
        he += [b for _a, b in c["hp"]]
        ns += [a for a, _b in c["nonhp"]]
        ne += [b for _a, b in c["nonhp"]]
        hb.append(len(hs))
        nb.append(len(ns))
        lens.append(c["len_read"])
    return chunks.ChunkTable(lens, hb, hs, he, nb, ns, ne)


def test_native_split_of_a_listing_block_writes_the_same_files_as_the_python_loop(golden, tmp_path):
    """``cf_listing_split_npy_int16`` (csrc/split_host.hpp; host code of the C-ABI library, no GPU) over a directory holding the golden
    cases as ``.npy`` reads -- edge slices, several dots, an empty stem --, plus reads WITHOUT homopolymer rows (skipped, like the
    reference's ``for read in hp_dict``) that are not even int16 vectors (never opened): same names, same bytes as ``split_reads``'
    Python loop and as the reference-executed goldens; a block [lo, hi) inside the listing; counts."""
    from catfish_amd import sharding
    doc, arrays = golden
    reads = tmp_path / "reads"
    reads.mkdir()
    cases = []
    for i, c in enumerate(doc):
        if not c["name"].endswith(".fast5"):
            continue                                               # (a name without extension is not a .npy read: general loop only)
        name = _npy(c["name"])
        np.save(reads / name, arrays["signal_%d" % i])
        cases.append(dict(c, name=name, index=i))
    (reads / "zz_no_hp.npy").write_bytes(b"not a numpy file at all")          # no HP rows: never read
    cases.append({"name": "zz_no_hp.npy", "len_read": 10, "hp": [], "nonhp": [[0, 10]], "index": None})
    np.save(reads / "aa_no_hp.npy", np.arange(6, dtype=np.int32))
    cases.append({"name": "aa_no_hp.npy", "len_read": 6, "hp": [], "nonhp": [[0, 6]], "index": None})
    cases.sort(key=lambda c: os.fsencode(c["name"]))
    listing = sharding.DirListing(str(reads))
    assert listing.names() == [c["name"] for c in cases]
    table = _table_from_cases(cases)
    paths = [str(reads / c["name"]) for c in cases]
    for d in ("n/HP", "n/nonHP", "p/HP", "p/nonHP", "b/HP", "b/nonHP"):
        (tmp_path / d).mkdir(parents=True)
    native = split.split_listing(table, listing, 0, str(tmp_path / "n/HP"), str(tmp_path / "n/nonHP"), n_threads=3)
    python = split.split_reads(table, paths, str(tmp_path / "p/HP"), str(tmp_path / "p/nonHP"))
    assert native == python and native["reads"] == len(cases) - 2
    for d in ("HP", "nonHP"):
        names = sorted(os.listdir(tmp_path / "n" / d))
        assert names == sorted(os.listdir(tmp_path / "p" / d)) and names
        assert all((tmp_path / "n" / d / f).read_bytes() == (tmp_path / "p" / d / f).read_bytes() for f in names)
    for c in cases:                                                # ... and the goldens themselves
        for j, (folder, name) in enumerate(doc[c["index"]]["files"] if c["index"] is not None else []):
            assert (tmp_path / "n" / folder / _npy(name)).read_bytes() == _saved(arrays["out_%d_%d" % (c["index"], j)])
    # a block inside the listing: rows of the table belong to entries [lo, lo + len)
    lo, hi = 3, 9
    sub = table.take(np.arange(lo, hi))
    part = split.split_reads(sub, paths[lo:hi], str(tmp_path / "b/HP"), str(tmp_path / "b/nonHP"), listing=listing, lo=lo)
    assert part["reads"] == sum(1 for c in cases[lo:hi] if c["hp"])
    stems = {c["name"].split(".")[0] for c in cases[lo:hi] if c["hp"]}
    assert {f.rsplit("_", 1)[0] for f in os.listdir(tmp_path / "b/HP")} == stems
    assert all((tmp_path / "b/HP" / f).read_bytes() == (tmp_path / "n/HP" / f).read_bytes() for f in os.listdir(tmp_path / "b/HP"))
    listing.close()


def test_native_split_refuses_what_it_cannot_read_and_reports_io_errors(tmp_path):
    """A read WITH homopolymer rows that is not an int16 ``.npy`` vector: ValueError naming it from ``split_listing``, and
    ``split_reads`` then takes the whole block through the general loop (which reads int32 codes, ``.npz`` and ``.bin`` too).  An output
    directory that does not exist: OSError with the system's reason (an exception aborts the run, split_f5.py:23-32)."""
    from catfish_amd import sharding
    reads = tmp_path / "reads"
    reads.mkdir()
    sig = np.arange(100, dtype=np.int16)
    np.save(reads / "a.npy", sig)
    np.save(reads / "b.npy", sig.astype(np.int32))                  # codes in int32: infer.load_dac reads them, the native pool does not
    np.savez(reads / "c.npz", raw=sig)
    listing = sharding.DirListing(str(reads))
    table = _table_from_cases([{"len_read": 100, "hp": [[0, 40]], "nonhp": [[40, 100]]}] * 3)
    paths = [str(reads / n) for n in ("a.npy", "b.npy", "c.npz")]
    for d in ("HP", "nonHP"):
        (tmp_path / d).mkdir()
    with pytest.raises(ValueError, match="c.npz"):
        split.split_listing(table, listing, 0, str(tmp_path / "HP"), str(tmp_path / "nonHP"))
    two = table.take(np.arange(2))
    with pytest.raises(ValueError, match="b.npy"):
        split.split_listing(two, listing, 0, str(tmp_path / "HP"), str(tmp_path / "nonHP"))
    done = split.split_reads(table, paths, str(tmp_path / "HP"), str(tmp_path / "nonHP"), listing=listing, lo=0)
    assert done == {"reads": 3, "files_hp": 3, "files_nonhp": 3, "samples": 300}
    for stem in "abc":
        assert np.array_equal(np.load(tmp_path / "HP" / (stem + "_0.npy")), sig[:40])
        assert np.array_equal(np.load(tmp_path / "nonHP" / (stem + "_1.npy")), sig[40:])
    one = table.take(np.arange(1))
    with pytest.raises(OSError, match="nowhere"):
        split.split_listing(one, listing, 0, str(tmp_path / "nowhere"), str(tmp_path / "nonHP"))
    (tmp_path / "HP" / "a_0.npy").unlink()
    (tmp_path / "HP" / "a_0.npy").mkdir()                           # the piece's name is taken by a directory: open() fails
    with pytest.raises(OSError, match="a_0.npy"):
        split.split_listing(one, listing, 0, str(tmp_path / "HP"), str(tmp_path / "nonHP"))
    listing.close()
