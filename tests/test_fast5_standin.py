"""The FAST5 branch of the read loader against a stand-in ``h5py``.

The reference opens every read with ``h5py.File`` (catfish/infer.py:25-29), trims the leader at ``first_sample_template`` and
finds the single read's group through ``visit(str)`` (infer.py:87-90).  h5py / libhdf5 are in neither container and the
reference ships no sample file, so that branch of ``catfish_amd.infer`` (``load_dac`` -> ``_trimmed_fast5_signal``,
``process_signal``) cannot run against the real library here.  This test puts a minimal ``h5py`` module in ``sys.modules`` -- a
``File`` context manager over nested groups with ``.attrs``, ``visit`` and dataset ``[()]``, the four things the reference's code
touches -- and drives the branch end to end.  It proves the branch is not dead code and follows the reference's lookups; it is
NOT a test of HDF5 decoding (out of scope: SURVEY.md section 2, #7)."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from catfish_amd import infer  # noqa: E402
from oracle import catfish_oracle as oracle  # noqa: E402


class _Dataset(object):
    def __init__(self, array):
        self._array = np.asarray(array)

    def __getitem__(self, key):
        if key != ():
            raise TypeError("the reference reads whole datasets: [()]")
        return self._array.copy()


class _Group(object):
    """An HDF5 group: children by name, ``attrs``, path look-ups with or without a trailing slash, ``visit``."""

    def __init__(self, children=None, attrs=None):
        self.children = dict(children or {})
        self.attrs = dict(attrs or {})

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, _Group) or part not in node.children:
                raise KeyError("Unable to open object (object '%s' doesn't exist)" % part)
            node = node.children[part]
        return node

    def visit(self, func):
        """h5py semantics: ``func(name)`` for every member below this group, depth first in name order; the first return value
        that is not None stops the walk and is returned (``visit(str)`` therefore yields the first member's name)."""
        def walk(group, prefix):
            for name in sorted(group.children):
                full = prefix + name
                got = func(full)
                if got is not None:
                    return got
                child = group.children[name]
                if isinstance(child, _Group):
                    got = walk(child, full + "/")
                    if got is not None:
                        return got
            return None
        return walk(self, "")


def _fake_h5py(registry, opened):
    """A module object with ``File(path, mode)``; the tree of a path comes from ``registry``."""
    mod = types.ModuleType("h5py")

    class File(_Group):
        def __init__(self, path, mode="r"):
            if mode != "r":
                raise ValueError("the reference opens reads read-only")
            if path not in registry:
                raise OSError("Unable to open file (file signature not found)")
            tree = registry[path]
            _Group.__init__(self, tree.children, tree.attrs)
            self.closed = False
            opened.append(self)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.closed = True
            return False

    mod.File = File
    return mod


def _single_read_tree(signal, first_sample, read_name="Read_1042"):
    return _Group({
        "Analyses": _Group({"Segmentation_000": _Group({"Summary": _Group({
            "segmentation": _Group(attrs={"first_sample_template": first_sample, "duration_template": len(signal) - first_sample})})})}),
        "Raw": _Group({"Reads": _Group({read_name: _Group({"Signal": _Dataset(signal)}, attrs={"read_number": 1042})})}),
        "UniqueGlobalKey": _Group({"channel_id": _Group(attrs={"offset": 10.0, "range": 1400.0, "digitisation": 8192.0})}),
    })


class _ScoreModel(object):
    """Stands in for the network: scores from the windows it is handed (a smoothed sigmoid of the signal), so that the spans
    depend on the trimmed, normalised, padded input and on nothing else."""

    def __init__(self):
        self.seen = None

    def infer(self, raw_in):
        self.seen = np.array(raw_in, dtype=np.float64)
        flat = self.seen.reshape(-1)
        smooth = np.convolve(flat, np.ones(41) / 41.0, mode="same")
        return 1.0 / (1.0 + np.exp(-4.0 * smooth))


@pytest.fixture
def fast5(tmp_path, monkeypatch):
    rng = np.random.default_rng(5)
    # a squiggle whose level wanders (so that the smoothed score crosses 0.5 in long runs) after a leader of 300 samples
    body = oracle.synthetic_dac(1, 3000, seed=12)[0].astype(np.int64) + np.repeat(rng.integers(-90, 90, size=30), 100)
    signal = np.concatenate([rng.integers(300, 700, size=300), body]).astype(np.int16)
    path = str(tmp_path / "read_ch12.fast5")
    open(path, "wb").write(b"\x89HDF\r\n\x1a\n stand-in, never parsed")      # os.path.exists() must hold (infer.py:25)
    registry, opened = {path: _single_read_tree(signal, 300)}, []
    monkeypatch.setitem(sys.modules, "h5py", _fake_h5py(registry, opened))
    return {"path": path, "signal": signal, "first": 300, "opened": opened, "registry": registry, "tmp": tmp_path}


def test_load_dac_trims_the_leader_of_the_single_read(fast5):
    got = infer.load_dac(fast5["path"])
    assert got.dtype == np.int16 and got.ndim == 1
    assert np.array_equal(got, fast5["signal"][fast5["first"]:])                  # infer.py:87-90
    assert len(fast5["opened"]) == 1 and fast5["opened"][0].closed                  # opened once, closed by the with block (infer.py:27)
    assert infer.is_dac(got)


def test_process_signal_on_an_open_file_matches_the_reference_rule(fast5):
    import h5py
    with h5py.File(fast5["path"], "r") as fh:
        got = infer.process_signal(fh)                                             # infer.py:77-93: an OPEN file, as in the reference
    want = oracle.normalize_raw_signal(fast5["signal"][fast5["first"]:])
    assert got.dtype == np.float64 and np.array_equal(got, want)
    with h5py.File(fast5["path"], "r") as fh, pytest.raises(ValueError, match="norm_method not recognized"):
        infer.process_signal(fh, normalization="zscore")                           # infer.py:104


def test_visit_str_finds_the_read_whatever_it_is_called(fast5):
    fast5["registry"][fast5["path"]] = _single_read_tree(fast5["signal"], 17, read_name="Read_7")
    assert np.array_equal(infer.load_dac(fast5["path"]), fast5["signal"][17:])
    # first_sample_template 0: nothing trimmed; numpy integer attribute, as h5py returns it
    fast5["registry"][fast5["path"]] = _single_read_tree(fast5["signal"], np.int64(0))
    assert np.array_equal(infer.load_dac(fast5["path"]), fast5["signal"])


def test_infer_class_from_signal_through_the_fast5_branch(fast5):
    model = _ScoreModel()
    spans, n = infer.infer_class_from_signal(fast5["path"], model)
    trimmed = fast5["signal"][fast5["first"]:]
    assert n == len(trimmed) == 3000
    # what the model was handed: normalised, zero-padded to whole windows, [N, 35, 1] (infer.py:31-43)
    want_in, pad = oracle.pad_and_window(oracle.normalize_raw_signal(trimmed))
    assert model.seen.shape == (len(trimmed) // 35 + 1, 35, 1) and pad == 35 - len(trimmed) % 35
    assert np.array_equal(model.seen.reshape(-1), np.asarray(want_in, dtype=np.float64).reshape(-1))
    # what came back: the reference's tail on the model's scores (infer.py:46-51), restated by the oracle
    scores = _ScoreModel().infer(model.seen)[:-pad]
    want_spans = oracle.hp_in_pred(oracle.correct_short(oracle.class_from_threshold(scores)))
    assert spans == [list(s) for s in want_spans] and len(spans) >= 1


def test_wrong_paths_and_broken_files_fail_like_the_reference(fast5):
    with pytest.raises(ValueError, match="path to FAST5 is not correct"):         # infer.py:25-26
        infer.infer_class_from_signal(str(fast5["tmp"] / "nope.fast5"), _ScoreModel())
    with pytest.raises(ValueError, match="path to FAST5 is not correct"):
        infer.load_dac(str(fast5["tmp"] / "nope.fast5"))
    # a FAST5 without the segmentation analysis: the reference lets h5py's KeyError through (infer.py:87)
    tree = _single_read_tree(fast5["signal"], 5)
    del tree.children["Analyses"]
    fast5["registry"][fast5["path"]] = tree
    with pytest.raises(KeyError):
        infer.load_dac(fast5["path"])
    # a file h5py cannot open (not in the registry = "file signature not found")
    other = str(fast5["tmp"] / "garbage.fast5")
    open(other, "wb").write(b"junk")
    with pytest.raises(OSError):
        infer.load_dac(other)


def test_without_h5py_the_branch_says_what_is_missing(fast5, monkeypatch):
    monkeypatch.setitem(sys.modules, "h5py", None)                                  # import h5py -> ImportError
    with pytest.raises(ImportError, match="needs h5py"):
        infer.load_dac(fast5["path"])


def test_converter_writes_the_reads_the_native_loader_takes(fast5, tmp_path):
    """``python -m catfish_amd.convert``: FAST5 (through the stand-in h5py) -> int16 ``.npy`` holding exactly ``load_dac``'s samples,
    byte for byte ``numpy.save``; the result goes through the byte-level reader and the library's loader pool; an unreadable file
    stops the run (or is reported with ``--keep-going``)."""
    import io
    from catfish_amd import convert
    src_dir = os.path.dirname(fast5["path"])
    out = tmp_path / "npy"
    done = convert.convert_directory(src_dir, str(out))
    assert done == {"converted": 1, "samples": 3000, "failed": []}
    dst = out / "read_ch12.npy"
    want = fast5["signal"][fast5["first"]:]
    buf = io.BytesIO()
    np.save(buf, want)
    assert dst.read_bytes() == buf.getvalue() and not (out / "read_ch12.npy.part").exists()
    assert np.array_equal(infer._read_npy_int16(str(dst)), want)                   # the fast path of infer.load_dac takes it
    from catfish_amd import sharding
    listing = sharding.DirListing(str(out))
    assert listing.names() == ["read_ch12.npy"] and listing.sizes(0, 1).tolist() == [len(buf.getvalue())]
    listing.close()
    open(os.path.join(src_dir, "zz_garbage.fast5"), "wb").write(b"junk")            # not in the stand-in's registry: h5py cannot open it
    with pytest.raises(OSError):
        convert.convert_directory(src_dir, str(out))
    assert convert.main(["-i", src_dir, "-o", str(out), "--keep-going"]) == 1
    kept = convert.convert_directory(src_dir, str(out), keep_going=True)
    assert kept["converted"] == 1 and [n for n, _e in kept["failed"]] == ["zz_garbage.fast5"]
    # samples that are not int16 codes are refused, not truncated
    fast5["registry"][fast5["path"]] = _single_read_tree(np.array([0.5, 1.5, 2.5, 70000.0]), 0)
    with pytest.raises(ValueError, match="int16 DAC codes"):
        convert.convert_read(fast5["path"], str(out / "x.npy"))
