"""Host-side logic of the product (no GPU): vectorised post-processing, packing, sharding, CLI tail,
hyper-parameter parser, operator-surface error behaviour, C-ABI symbol table."""
import json
import os
import subprocess

import numpy as np
import pytest

from catfish_amd import batching, cli, infer, metrics, neural_network, sharding
from catfish_amd import _native
from oracle import catfish_oracle as oracle
from conftest import GOLDEN, REFERENCE, ROOT, has_reference


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "postproc_golden.json")) as fh:
        return json.load(fh)


def test_postprocessing_matches_reference_goldens(golden):
    for c in golden["postproc"]:
        labels = infer.class_from_threshold(c["scores"])
        assert labels == c["labels"]
        corrected = infer.correct_short(labels)
        assert corrected.tolist() == c["corrected"]
        assert infer.hp_in_pred(corrected) == c["spans"]
        assert all(isinstance(v, int) for s in infer.hp_in_pred(corrected) for v in s)


def test_postprocessing_matches_oracle_on_random_inputs():
    rng = np.random.default_rng(3)
    for n in (1, 2, 29, 30, 31, 500, 5000):
        p = rng.random(n) < rng.choice([0.2, 0.5, 0.8, 0.95])
        lab = p.astype(int)
        assert np.array_equal(infer.correct_short(lab), oracle.correct_short(lab))
        assert infer.hp_in_pred(lab) == oracle.hp_in_pred(lab)
        assert infer.hp_in_pred(lab, 0, 0, label=0) == oracle.hp_in_pred(lab, 0, 0, label=0)


def test_postprocessing_empty_raises_like_reference():
    with pytest.raises(IndexError):
        infer.correct_short([])
    with pytest.raises(IndexError):
        infer.hp_in_pred([])


def test_normalisation_and_padding(golden):
    for c in golden["normalize"]:
        out = infer.normalize_raw_signal(np.array(c["raw"], dtype=np.int16), "median")
        assert np.array_equal(out, np.array(c["out"]))
    with pytest.raises(ValueError):
        infer.normalize_raw_signal(np.arange(4), "mean")
    for c in golden["padding"]:
        assert infer.padding_size_for(c["length"]) == c["padding_size"]
    with pytest.raises(ValueError):
        infer.load_raw("/nonexistent/read.fast5")


def test_cli_tail_matches_reference(golden):
    for c in golden["center_hp"]:
        out = cli.center_hp([[0, 5], list(c["in"])], c["len_read"], c["chunk_size"])
        assert out[-1] == c["out"]
    for c in golden["merge"]:
        merged = cli.merge_positions([list(s) for s in c["spans"]], c["len_read"], c["chunk_size"])
        assert merged == c["merged"]
        assert cli.nonhp_complement(merged, c["len_read"]) == c["nonhp"]


def test_native_chunk_tables_follow_the_reference_rules(golden):
    """chunks.ChunkTable.from_spans (cf_chunks_from_spans: host code of the C-ABI library, no GPU needed) = the reference's
    merge / center_hp / complement loop (catfish/catfish:57-82,121-135) for every read of a table: against the goldens made
    from the reference's own functions, and against the per-read Python rules on random reads -- incl. reads whose FIRST span
    is longer than a chunk (the merged list then holds the same list object twice and `hp_positions[i - 1]` is the LAST span),
    reads without spans (the odd `[([(0, len), len])]` entry) and reads that their chunks cover completely (`[]`)."""
    import copy
    import json
    from catfish_amd import chunks
    # 1. the reference-executed goldens, one table per chunk size
    for size in sorted({c["chunk_size"] for c in golden["merge"]}):
        cases = [c for c in golden["merge"] if c["chunk_size"] == size]
        bounds = np.concatenate(([0], np.cumsum([len(c["spans"]) for c in cases])))
        flat = np.array([sp for c in cases for sp in c["spans"]], dtype=np.int64).reshape(-1, 2)
        tab = chunks.ChunkTable.from_spans(bounds, flat[:, 0], flat[:, 1], [c["len_read"] for c in cases], size)
        hp, non = tab.to_dicts(list(range(len(cases))))
        for i, c in enumerate(cases):
            assert hp[i] == c["merged"] and non[i] == c["nonhp"]
    # 2. random reads against the Python rules, through the JSON text
    rng = np.random.default_rng(0)
    reads = []
    for _ in range(1500):
        length = int(rng.integers(20, 9000))
        spans, pos = [], int(rng.integers(-11, 200))
        while rng.random() > 0.1:
            n = int(rng.choice([15, 20, 40, 100, 600, 1200, 2500])) + 27
            if pos + n > length + 16:
                break
            spans.append([pos, pos + n])
            pos += n + int(rng.integers(1, 900))
        reads.append((spans, length))
    assert sum(1 for sp, _n in reads if sp and sp[0][1] - sp[0][0] >= 1000) > 50 and sum(1 for sp, _n in reads if not sp) > 50
    names = ["read_%d.fast5" % i for i in range(len(reads))]
    names[5] = 'we"ird\\name\u00e9.fast5'
    want_hp, want_non = {}, {}
    for name, (spans, length) in zip(names, reads):
        merged, non = cli.chunks_of_read(copy.deepcopy(spans), length, 1000)
        if merged is not None:
            want_hp[name] = merged
        want_non[name] = non
    assert any(v == [] for v in want_non.values())
    bounds = np.concatenate(([0], np.cumsum([len(sp) for sp, _n in reads])))
    flat = np.array([p for sp, _n in reads for p in sp], dtype=np.int64).reshape(-1, 2)
    tab = chunks.ChunkTable.from_spans(bounds, flat[:, 0], flat[:, 1], [n for _sp, n in reads], 1000)
    for plain in (False, True):                                # escaped keys, then the no-escaping fast path
        keys = ["r%d" % i for i in range(len(names))] if plain else names
        hp_text, non_text = tab.json_members(keys)
        ren = dict(zip(names, keys))
        assert b"{" + hp_text + b"}" == json.dumps({ren[k]: v for k, v in want_hp.items()}).encode()
        assert b"{" + non_text + b"}" == json.dumps({ren[k]: v for k, v in want_non.items()}).encode()
    got_hp, got_non = tab.to_dicts(names)
    assert got_hp == want_hp and json.dumps(got_non) == json.dumps(want_non)
    # 3. tables concatenate and re-order like lists of reads
    cut = [0, 1, 700, 700, 1500]
    parts = [chunks.ChunkTable.from_spans(bounds[a:b + 1], flat[:, 0], flat[:, 1], [n for _sp, n in reads[a:b]], 1000)
             for a, b in zip(cut[:-1], cut[1:])]
    assert chunks.ChunkTable.concat(parts).json_members(names) == tab.json_members(names)
    order = rng.permutation(len(reads))
    assert tab.take(order).json_members([names[i] for i in order]) == \
        chunks.ChunkTable.from_spans(*_reordered(bounds, flat, reads, order), 1000).json_members([names[i] for i in order])
    empty = chunks.ChunkTable.concat([])
    assert len(empty) == 0 and empty.json_members([]) == (b"", b"") and empty.to_dicts([]) == ({}, {})
    with pytest.raises(ValueError):
        chunks.ChunkTable.from_spans([0, 1], [5], [9], [100, 200], 1000)


def _reordered(bounds, flat, reads, order):
    rows = np.concatenate([np.arange(bounds[i], bounds[i + 1]) for i in order]).astype(np.int64)
    nb = np.concatenate(([0], np.cumsum([bounds[i + 1] - bounds[i] for i in order])))
    return nb, flat[rows, 0], flat[rows, 1], [reads[i][1] for i in order]


def test_cli_options_match_reference():
    main = cli._build_click_main()
    opts = {o.name: o for o in main.params}
    # the reference's three options, unchanged; --gpus / --network-path / --precision are MI355X additions
    assert {"input_dir", "split_dir", "chunk_size"} <= set(opts)
    assert set(opts) - {"input_dir", "split_dir", "chunk_size"} == {"gpus", "network_path", "precision"}
    assert opts["gpus"].default == 1 and opts["network_path"].default == "ResNetRNN" and opts["precision"].default == "fp32"
    assert opts["input_dir"].opts == ["--input-dir", "-i"]
    assert opts["split_dir"].opts == ["--split-dir", "-s"]
    assert opts["chunk_size"].opts == ["--chunk-size", "-c"] and opts["chunk_size"].default == 1000


def test_pack_reads_layout():
    rng = np.random.default_rng(0)
    lens = [1, 34, 35, 36, 70, 512, 4096]
    sigs = [rng.normal(size=n) for n in lens]
    pk = batching.pack_reads(sigs)
    want_win = [oracle.pad_and_window(s)[0].shape[0] for s in sigs]
    assert np.array_equal(np.diff(pk.win_offsets), want_win)
    for i, s in enumerate(sigs):
        x, pad = oracle.pad_and_window(s)
        got = pk.x[pk.win_offsets[i]:pk.win_offsets[i + 1]]
        assert np.array_equal(got, x[:, :, 0].astype(np.float32))
    assert pk.n_windows == sum(want_win) and pk.n_reads == len(lens)


def test_length_buckets_cover_all_reads_once():
    rng = np.random.default_rng(2)
    lens = np.exp(rng.uniform(np.log(512), np.log(16384), size=300)).astype(int)
    buckets = batching.length_buckets(lens, max_windows=4096)
    flat = sorted(i for b in buckets for i in b)
    assert flat == list(range(300))
    for b in buckets:
        w = sum(sharding.windows_of(lens[i]) for i in b)
        assert w <= 4096 or len(b) == 1


def test_spans_from_packed_labels_match_per_read_reference_logic():
    rng = np.random.default_rng(5)
    lens = [40, 35, 700, 123, 70]
    sigs = [np.zeros(n) for n in lens]
    pk = batching.pack_reads(sigs)
    labels = np.zeros(pk.n_windows * 35, dtype=np.uint8)
    per_read = []
    for i, n in enumerate(lens):
        lab = (rng.random(n) < 0.7).astype(int)
        lab = oracle.correct_short(lab)
        per_read.append(lab)
        labels[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = lab
    got = batching.spans_from_labels(labels, pk.sample_offsets, pk.n_reads)
    for i, lab in enumerate(per_read):
        want = oracle.hp_in_pred(lab) if lab.any() else []
        assert got[i] == want


def test_spans_from_runs_equals_spans_from_labels():
    rng = np.random.default_rng(6)
    lens = [40, 35, 700, 123, 70, 2000]
    pk = batching.pack_reads([np.zeros(n) for n in lens])
    labels = np.zeros(pk.n_windows * 35, dtype=np.uint8)
    for i, n in enumerate(lens):
        labels[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = oracle.correct_short((rng.random(n) < 0.8).astype(int))
    d = np.diff(np.concatenate(([0], labels.astype(np.int8), [0])))
    starts, ends = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
    assert batching.spans_from_runs(starts, ends, pk.sample_offsets, pk.n_reads) == \
        batching.spans_from_labels(labels, pk.sample_offsets, pk.n_reads)


def test_shard_reads_balanced_and_complete():
    rng = np.random.default_rng(7)
    lens = np.exp(rng.uniform(np.log(512), np.log(16384), size=1000)).astype(int)
    for world in (1, 2, 4, 8):
        shards = sharding.shard_reads(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(1000))
        loads = [sum(sharding.windows_of(lens[i]) for i in s) for s in shards]
        assert max(loads) - min(loads) <= sharding.windows_of(16384)
    eq = sharding.shard_reads([4096] * 100000, 8)
    assert [len(s) for s in eq] == [12500] * 8


def test_confusion_matrix():
    assert metrics.confusion_matrix([1, 0, 1, 0], [1, 1, 0, 0]) == (1, 1, 1, 1)
    with pytest.raises(ValueError):
        metrics.confusion_matrix([1], [1, 0])


def test_hyperparams_parser(tmp_path):
    f = tmp_path / "ResNetRNN.txt"
    f.write_text("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\nlearning_rate: 0.001\n"
                 "layer_size: 64\nn_layers: 3\nkeep_prob: 0.8\nlayer_size_res: 32\nn_layers_res: 2\n\n"
                 "NEXT EPOCH\n\tAccuracy: 88.67%\n")
    hp = neural_network.retrieve_hyperparams(str(f))
    assert hp == dict(batch_size=256, optimizer_choice="RMSProp", learning_rate=0.001, layer_size=64,
                      n_layers=3, keep_prob=0.8, layer_size_res=32, n_layers_res=2)
    if has_reference():
        assert neural_network.retrieve_hyperparams(os.path.join(REFERENCE, "catfish/ResNetRNN/ResNetRNN.txt")) == hp


def test_model_surface_and_errors(hp):
    from catfish_amd.resnet_class import ResNetRNN
    from catfish_amd.rnn_class import RNN
    m = ResNetRNN(**hp)
    assert (m.window, m.n_inputs, m.n_outputs, m.batch_size) == (35, 1, 1, 256)
    assert m.model_type == "ResNet-RNN" and (m.tp, m.fp, m.tn, m.fn) == (0, 0, 0, 0)
    assert RNN(**hp).model_type == "biGRU-RNN"
    bad = dict(hp, optimizer_choice="SGD")
    with pytest.raises(ValueError, match="optimizer"):
        ResNetRNN(**bad)
    with pytest.raises(RuntimeError):
        m.infer(np.zeros((1, 35, 1)))
    with pytest.raises(KeyError):
        ResNetRNN(batch_size=1)
    w = m._initial_weights(seed=0)
    assert len(w) == 74 and w["conv1d_2/kernel"].shape == (3, 32, 32)
    assert np.all(w["stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/gru_cell/gates/bias"] == 1)


def test_c_abi_exports_every_declared_symbol():
    """The library loads and exports exactly what include/catfish_hip.h declares (no compute calls)."""
    import re
    header = open(os.path.join(ROOT, "include", "catfish_hip.h")).read()
    declared = set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.SYMBOLS)
    assert os.path.exists(_native.LIB_PATH), "run `python -m catfish_amd.build` first"
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], stdout=subprocess.PIPE,
                         universal_newlines=True, check=True).stdout
    exported = set(re.findall(r" T (cf_[a-z0-9_]+)", out))
    assert declared <= exported
    lib = _native.lib()
    assert b"gfx950" in lib.cf_version()
    assert lib.cf_profile_slot_name(3) == b"gru_layer_mid" and lib.cf_profile_slot_name(8) == b"gru_fused"


def test_product_never_imports_oracle():
    import re
    pkg = os.path.join(ROOT, "catfish_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), name


def test_quiet_gc_restores_the_collector():
    """batching.quiet_gc: the cyclic collector is off inside (list building stays cheap next to torch's object graph) and
    back to its previous state afterwards, also when it was off before and when the body raises."""
    import gc
    was = gc.isenabled()
    try:
        gc.enable()
        with batching.quiet_gc():
            assert not gc.isenabled()
        assert gc.isenabled()
        with pytest.raises(RuntimeError):
            with batching.quiet_gc():
                raise RuntimeError("boom")
        assert gc.isenabled()
        gc.disable()
        with batching.quiet_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable() if was else gc.disable()


def test_fast_npy_reader_matches_numpy(tmp_path):
    """infer.load_dac's one-read() fast path for 1-D little-endian int16 .npy files (what a directory of DAC reads holds)
    returns exactly np.load's array; every other layout (other dtype, 2-D, big-endian, Fortran order, empty, truncated)
    takes the np.load route and gives what np.load gives."""
    from catfish_amd import infer
    rng = np.random.default_rng(3)
    for n in (1, 35, 4096, 70001):
        a = rng.integers(-32768, 32767, size=n).astype(np.int16)
        p = str(tmp_path / ("r%d.npy" % n))
        np.save(p, a)
        assert infer._read_npy_int16(p) is not None
        got = infer.load_dac(p)
        assert got.dtype == np.int16 and np.array_equal(got, a)
    odd = [np.arange(10, dtype=np.float32), np.arange(12, dtype=np.int16).reshape(3, 4), np.arange(5, dtype=">i2"),
           np.asfortranarray(np.arange(6, dtype=np.int16).reshape(2, 3)), np.arange(9, dtype=np.int32)]
    for i, a in enumerate(odd):
        p = str(tmp_path / ("odd%d.npy" % i))
        np.save(p, a)
        assert infer._read_npy_int16(p) is None
        assert np.array_equal(infer.load_dac(p), np.asarray(a).reshape(-1))
    p = str(tmp_path / "trunc.npy")
    np.save(p, np.arange(100, dtype=np.int16))
    with open(p, "rb") as fh:
        buf = fh.read()
    with open(p, "wb") as fh:
        fh.write(buf[:-10])
    assert infer._read_npy_int16(p) is None            # the length check sends it to np.load, which reports the damage
    with pytest.raises(Exception):
        infer.load_dac(p)


def test_missing_library_fails_loudly(hp):
    """No CPU fallback: with the HIP library absent the product raises NativeLibraryMissing at the first use (model
    restore, engine construction) -- it never routes around the kernels."""
    import subprocess
    import sys
    code = ("import os, sys\n"
            "from catfish_amd import _native, neural_network\n"
            "try:\n"
            "    _native.lib()\n"
            "except _native.NativeLibraryMissing as e:\n"
            "    print('LIB', type(e).__name__)\n"
            "m = neural_network.build_model('ResNetRNN', **%r)\n"
            "try:\n"
            "    m.initialize_network(seed=0)\n"
            "except _native.NativeLibraryMissing as e:\n"
            "    print('MODEL', type(e).__name__)\n" % (dict(hp),))
    env = dict(os.environ, CATFISH_HIP_LIB="/nonexistent/libcatfish_hip.so", CATFISH_DEBUG_KNOBS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "LIB NativeLibraryMissing" in out.stdout and "MODEL NativeLibraryMissing" in out.stdout


def test_library_override_needs_the_debug_switch_and_a_matching_abi(tmp_path):
    """``CATFISH_HIP_LIB`` is a debug knob: ignored unless ``CATFISH_DEBUG_KNOBS=1`` (a stray variable must not swap the
    product library), named on stderr when it takes effect, and a library built for another ``CF_ABI_VERSION`` is refused
    instead of being called with the wrong arguments."""
    import shutil
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    src = tmp_path / "stale.c"
    src.write_text("int cf_abi_version(void) { return 2; }\n")
    stale = tmp_path / "libstale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(stale), str(src)], check=True)
    code = ("from catfish_amd import _native\n"
            "try:\n"
            "    _native.lib()\n"
            "    print('LOADED', _native.LIB_PATH)\n"
            "except _native.NativeLibraryMissing as e:\n"
            "    print('REFUSED', e)\n")
    base = {k: v for k, v in os.environ.items() if k not in ("CATFISH_DEBUG_KNOBS", "CATFISH_HIP_LIB")}
    quiet = subprocess.run([sys.executable, "-c", code], env=dict(base, CATFISH_HIP_LIB=str(stale)), capture_output=True,
                           text=True, cwd=ROOT, timeout=300)
    assert "LOADED " + _native.DEFAULT_LIB_PATH in quiet.stdout and "debug knob" not in quiet.stderr
    loud = subprocess.run([sys.executable, "-c", code], env=dict(base, CATFISH_HIP_LIB=str(stale), CATFISH_DEBUG_KNOBS="1"),
                          capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert "REFUSED" in loud.stdout and "ABI 2" in loud.stdout and "needs %d" % _native.CF_ABI_VERSION in loud.stdout
    assert "debug knob CATFISH_HIP_LIB" in loud.stderr


def test_roofline_is_reported_against_the_nearer_roof():
    """bench.nearer_roof: fp32 mid layer stays on the matrix roof, the bf16 one moves to HBM (by algorithmic bytes: input +
    output slab; the stored PMC traffic decides a near tie) and keeps the matrix view beside it."""
    import bench
    n = 256 * 4096
    fp32 = {"bound": "mfma", "achieved": 129.4, "peak": 157.3, "unit": "TFLOP/s", "frac": 129.4 / 157.3, "traffic": 1.586e9}
    out = bench.nearer_roof(dict(fp32), "fp32", n, 1.194e-3)
    assert out["bound"] == "mfma" and abs(out["hbm_view"]["frac"] - 1024 * n / 1.194e-3 / 8e12) < 1e-9
    assert abs(out["hbm_view"]["traffic_rate_frac"] - 1.586e9 / 1.194e-3 / 8e12) < 1e-9
    bf16 = {"bound": "mfma", "achieved": 867.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 867.0 / 2500.0, "traffic": 813.5e6}
    out = bench.nearer_roof(dict(bf16), "bf16", n, 178.3e-6)
    assert out["bound"] == "hbm" and out["unit"] == "GB/s" and out["peak"] == 8000.0
    assert abs(out["frac"] - 512 * n / 178.3e-6 / 8e12) < 1e-9 and out["mfma_view"]["frac"] == bf16["frac"]
    assert out["hbm_detail"]["traffic_rate_frac"] > out["frac"]            # 1.5x re-read: each direction reads the whole input
    tie = {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None}
    assert bench.nearer_roof(dict(tie), "bf16", n, 1e-3)["bound"] == "mfma"   # no traffic record, algorithmic bytes below
    # bf16x3 in profiles/r04_bench_default.json -- issued pipe 0.489, algorithmic HBM 0.354, MEASURED traffic 0.520 of
    # peak: the measured traffic takes part in the decision again, and a gap under 10 % is reported as a near tie
    secs = 0.37735e-3
    x3 = {"bound": "mfma", "achieved": 409.8, "peak": 2500.0, "unit": "TFLOP/s", "frac": 409.8 / 2500.0, "traffic": 0.520 * 8e12 * secs}
    out = bench.nearer_roof(dict(x3), "bf16x3", n, secs)
    assert out["bound"] == "hbm" and out["near_tie"] is True and "measured HBM traffic" in out["decided_by"] and "near tie" in out["decided_by"]
    assert abs(out["hbm_detail"]["traffic_rate_frac"] - 0.520) < 1e-6 and abs(out["matrix_pipe_issued"]["frac"] - 3 * 409.8 / 2500.0) < 1e-9
    assert abs(out["frac"] - 1024 * n / secs / 8e12) < 1e-9                  # frac / achieved stay algorithmic
    out = bench.nearer_roof(dict(x3, traffic=None), "bf16x3", n, secs)        # without the record: issued pipe 0.49 against 0.35
    assert out["bound"] == "mfma" and out["near_tie"] is False and "ISSUED matrix-pipe" in out["decided_by"]


def test_build_records_kernel_resources_and_no_kernel_spills():
    """gru_bf16x3_pipe_kernel is built around the whole 512-register file of a one-wave-per-SIMD launch; nothing
    recorded that its four instantiations compile without scratch.  catfish_amd/build.py now parses hipcc's
    -Rpass-analysis=kernel-resource-usage remarks into csrc/kernel_resources.json and refuses a build whose kernels spill."""
    import json
    from catfish_amd import build
    text = ("x.hip:9:1: remark: Function Name: _Z1kv [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     TotalSGPRs: 44 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     AGPRs: 124 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     ScratchSize [bytes/lane]: 16 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     Occupancy [waves/SIMD]: 1 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     VGPRs Spill: 4 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:9:1: remark:     LDS Size [bytes/block]: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    got = build.parse_resource_remarks(text)
    assert got == {"_Z1kv": {"sgprs": 44, "vgprs": 256, "agprs": 124, "scratch_bytes_per_lane": 16, "waves_per_simd": 1,
                             "vgpr_spills": 4, "lds_bytes": 0}}
    assert build.check_resources(got) and "scratch 16" in build.check_resources(got)[0]
    assert build.check_resources({"k": {"vgprs": 256, "agprs": 257, "scratch_bytes_per_lane": 0}})
    if not os.path.exists(build.RESOURCES):
        pytest.skip("library not built here")
    with open(build.RESOURCES) as fh:
        rec = json.load(fh)
    kernels = rec["kernels"]
    assert len(kernels) >= 70 and build.check_resources(kernels) == []
    x3 = {k: v for k, v in kernels.items() if "gru_bf16x3_pipe_kernel" in k}
    assert len(x3) == 4 and all(v["waves_per_simd"] == 1 and v["scratch_bytes_per_lane"] == 0 and 256 < v["vgprs"] + v["agprs"] <= 512
                                for v in x3.values())


def test_native_npy_loader_reads_what_the_python_loader_reads(tmp_path):
    """cf_load_npy_int16 (host code of the C-ABI library: a pool of threads reading int16 .npy reads straight into one
    buffer) against infer.load_dac on the same files; everything else is refused with the file named, so that the caller
    takes the general loader -- never a silent misread."""
    import ctypes as C
    lib = _native.lib()
    rng = np.random.default_rng(3)

    def call(paths, capacity, threads=4):
        enc = [os.fsencode(str(p)) for p in paths]
        blob = b"\x00".join(enc) + b"\x00"
        bounds = np.concatenate(([0], np.cumsum([len(e) + 1 for e in enc]))).astype(np.int64)
        out = np.full(max(capacity, 1), -7, dtype=np.int16)
        lengths = np.full(len(paths), -1, dtype=np.int64)
        total = C.c_int64(-1)
        rc = lib.cf_load_npy_int16(blob, bounds.ctypes.data_as(C.c_void_p), len(paths), out.ctypes.data_as(C.c_void_p), capacity,
                                   lengths.ctypes.data_as(C.c_void_p), C.byref(total), threads)
        return rc, out, lengths, int(total.value), lib.cf_last_error().decode()

    reads, paths = [], []
    for i, n in enumerate([4096, 0, 1, 35, 70000, 513] + [int(v) for v in rng.integers(1, 3000, size=60)]):
        r = rng.integers(-2000, 2047, size=n).astype(np.int16)
        p = tmp_path / ("r%03d.npy" % i)
        np.save(p, r)
        reads.append(r)
        paths.append(p)
    want = np.concatenate(reads)
    for threads in (1, 4, 0, 200):
        rc, out, lengths, total, _ = call(paths, len(want) + 10, threads)
        assert rc == 0 and total == len(want) and np.array_equal(lengths, [len(r) for r in reads])
        assert np.array_equal(out[:total], want) and (out[total:] == -7).all()
    for p, r in zip(paths, reads):
        assert np.array_equal(infer.load_dac(str(p)), r)
    # does not fit: refused, the needed size reported, nothing promised about the buffer
    rc, _out, _l, total, msg = call(paths, len(want) - 1)
    assert rc == _native.CF_ERR_INVALID and total == len(want) and "fit" in msg
    # files the general loader must take: other dtype, two dimensions, Fortran order, truncated, missing, not an .npy at all
    bad = {}
    np.save(tmp_path / "f32.npy", np.zeros(8, np.float32)); bad["f32.npy"] = True
    np.save(tmp_path / "i4.npy", np.zeros(8, np.int32)); bad["i4.npy"] = True
    np.save(tmp_path / "two_d.npy", np.zeros((4, 2), np.int16)); bad["two_d.npy"] = True
    np.save(tmp_path / "big_endian.npy", np.zeros(8, ">i2")); bad["big_endian.npy"] = True
    buf = open(paths[0], "rb").read()
    open(tmp_path / "short.npy", "wb").write(buf[:-2]); bad["short.npy"] = True
    open(tmp_path / "text.npy", "wb").write(b"hello"); bad["text.npy"] = True
    bad["missing.npy"] = True
    for name in bad:
        rc, _out, _l, _t, msg = call([paths[0], tmp_path / name, paths[2]], 100000)
        assert rc == _native.CF_ERR_INVALID and name in msg, name
    assert call([], 0)[0] == 0


def test_native_chunk_rules_property_based():
    """Property test (hypothesis) of cf_chunks_from_spans + cf_chunks_json against the per-read Python rules: arbitrary
    ascending span lists (overlapping, touching, negative starts, ends past the read), chunk sizes from 1 up, read lengths
    shorter than a chunk -- the integer quirks of catfish/catfish:57-82,121-135 must agree on every input, through the JSON
    text (i.e. including the `[([(0, len), len])]` and `[]` forms)."""
    import copy
    import json
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")
    from catfish_amd import chunks

    span = st.tuples(st.integers(-30, 6000), st.integers(1, 3000))
    read = st.tuples(st.lists(span, min_size=0, max_size=12), st.integers(1, 7000))

    @hyp.settings(max_examples=300, deadline=None)
    @hyp.given(st.lists(read, min_size=1, max_size=6), st.integers(1, 2500))
    def check(reads, chunk_size):
        fixed = []
        for spans, length in reads:
            starts = sorted(s for s, _n in spans)
            fixed.append(([[s, s + n] for s, (_s, n) in zip(starts, spans)], length))
        names = ["r%d" % i for i in range(len(fixed))]
        want_hp, want_non = {}, {}
        for name, (spans, length) in zip(names, fixed):
            merged, non = cli.chunks_of_read(copy.deepcopy(spans), length, chunk_size)
            if merged is not None:
                want_hp[name] = merged
            want_non[name] = non
        bounds = np.concatenate(([0], np.cumsum([len(sp) for sp, _n in fixed])))
        flat = np.array([p for sp, _n in fixed for p in sp], dtype=np.int64).reshape(-1, 2)
        tab = chunks.ChunkTable.from_spans(bounds, flat[:, 0], flat[:, 1], [n for _sp, n in fixed], chunk_size)
        hp_text, non_text = tab.json_members(names)
        assert b"{" + hp_text + b"}" == json.dumps(want_hp).encode()
        assert b"{" + non_text + b"}" == json.dumps(want_non).encode()

    check()


def test_bf16x3_schedule_compiles_on_the_host_and_is_consistent(tmp_path):
    """The instruction schedule of gru_bf16x3_pipe_kernel is plain constexpr C++ (catfish_amd/csrc/gru_bf16x3_sched.hpp): built
    here with g++ through tools/x3_sched_dump.cpp for all four kernel variants.  Every variant must pass its own
    program-order check (an MFMA never reads an operand that a later gap produces), schedule every micro-op exactly once,
    and -- for the 128-input layers, which are meant to hide their vector work -- load no gap with more than 44 issue cycles
    and no phase with more than 25 on average (24 fit behind an MFMA)."""
    import re
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ is not installed")
    exe = str(tmp_path / "x3dump")
    build = subprocess.run(["g++", "-std=c++17", "-DX3_SCHED_DUMP", "-o", exe, os.path.join(ROOT, "tools", "x3_sched_dump.cpp")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert build.returncode == 0, build.stdout
    for cin, last in ((128, False), (128, True), (32, False), (32, True)):
        args = [exe, str(cin)] + (["last"] if last else ["x"]) + ["-v"]
        out = subprocess.run(args, stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
        head = out.splitlines()[0]
        assert " ok 1," in head, head
        gaps = int(re.search(r"gaps (\d+)", head).group(1))
        assert gaps == (216 if cin == 128 else 108)
        ops = re.findall(r" ([A-Z]+[0-9]?[A-Z]?[0-9]?)(\d+)", "\n".join(l.split(":", 1)[1] for l in out.splitlines() if re.match(r"\s*\d+  p", l)))
        names = [a + b for a, b in ops]
        assert len(names) == len(set(names)), "a micro-op was scheduled twice"
        for kind, count in (("AE", 32), ("AR2", 32), ("AM", 32), ("AP4", 16), ("BE", 32), ("BR2", 32), ("CE", 32), ("CH", 32), ("CP4", 16),
                            ("LX", cin // 8), ("LB", 24), ("CL", 32 if last else 0), ("CS", 0 if last else 8)):
            assert sum(1 for n in names if re.fullmatch(kind + r"\d+", n)) == count, (cin, last, kind)
        if cin == 128:
            assert int(re.search(r"heaviest gap (\d+)", head).group(1)) <= 44
            for line in out.splitlines():
                m = re.search(r"= ([0-9.]+) per gap", line)
                if m:
                    assert float(m.group(1)) <= 25.0, line


def test_native_directory_listing_orders_like_sorted_and_serves_blocks(tmp_path):
    """sharding.DirListing (cf_listing_*: readdir + bytewise order by an 8-byte key behind the common prefix, strcmp on ties) against
    ``sorted(os.listdir())`` -- the order ``cli.run_pipeline`` used to build in Python (catfish/catfish:49-50 lists the directory;
    sorted, so that N ranks write the same bytes as one).  Names that tie in the key, names shorter than the common prefix + 8,
    a prefix-of-another name, non-ASCII UTF-8, a sub-directory, hidden files; sizes and names of blocks; the digest."""
    from catfish_amd import sharding
    d = tmp_path / "reads"
    d.mkdir()
    names = ["read_%05d.npy" % i for i in (3, 1, 2, 10, 100, 99999)] + ["read_", "read_1", "read_12345678", "read_123456789", "read_12345678a",
             "read_été.npy", "read_zz", "read_~", "read_A", "read_a", "read_0000000000000001.npy", "read_0000000000000002.npy"]
    for i, n in enumerate(names):
        (d / n).write_bytes(b"x" * (7 * i))
    (d / "read_subdir").mkdir()
    names.append("read_subdir")
    lst = sharding.DirListing(str(d))
    want = sorted(os.listdir(d))
    assert len(lst) == len(names) == len(want) and lst.names() == want
    assert lst.names(2, 5) == want[2:5] and lst.names(4, 4) == [] and lst.names(len(want) - 1) == want[-1:]
    sizes = lst.sizes(0, len(lst))
    assert [int(s) for s in sizes] == [os.stat(d / n).st_size for n in want]
    assert [int(s) for s in lst.sizes(3, 7, n_threads=3)] == [os.stat(d / n).st_size for n in want[3:7]] and len(lst.sizes(5, 5)) == 0
    with pytest.raises(ValueError):
        lst.sizes(3, len(lst) + 1)
    with pytest.raises(ValueError):
        lst.names(5, 4)
    # the same names -> the same digest; one name more, or one renamed -> another
    again = sharding.DirListing(str(d))
    assert again.digest == lst.digest and len(lst.digest) == 32
    (d / "read_00004.npy").write_bytes(b"")
    assert sharding.DirListing(str(d)).digest != lst.digest
    os.rename(d / "read_00004.npy", d / "read_00005.npy")
    more = sharding.DirListing(str(d))
    assert more.digest != lst.digest and more.names() == sorted(os.listdir(d))
    # a removed file shows when its size is asked for, by name
    os.unlink(d / "read_00005.npy")
    with pytest.raises(ValueError, match="read_00005.npy"):
        more.sizes(0, len(more))
    # no common prefix at all, an empty directory, a missing one
    e = tmp_path / "mixed"
    e.mkdir()
    for n in ("b", "a", "ab", "B", "0", "zz.fast5", ".hidden"):
        (e / n).write_bytes(b"1")
    assert sharding.DirListing(str(e)).names() == sorted(os.listdir(e))
    (tmp_path / "empty").mkdir()
    empty = sharding.DirListing(str(tmp_path / "empty"))
    assert len(empty) == 0 and empty.names() == [] and len(empty.sizes(0, 0)) == 0
    with pytest.raises(ValueError, match="cannot open directory"):
        sharding.DirListing(str(tmp_path / "nowhere"))
    # path strings for one block only
    paths = sharding.ListingPaths(lst).block(2, 6)
    assert len(paths) == len(lst) and paths[2] == str(d / want[2]) and paths[5] == str(d / want[5]) and paths[-1] == str(d / want[-1])
    assert paths[1:3] == [str(d / want[1]), str(d / want[2])]
    # what travels between ranks: the ordered names as one blob; a listing rebuilt from it is the same listing
    blob = lst.names_blob()
    twin = sharding.DirListing.from_names_blob(str(d), blob, len(lst))
    assert twin.digest == lst.digest and twin.names() == want and [int(x) for x in twin.sizes(0, len(twin))] == [int(x) for x in sizes]
    assert lst.names_blob(2, 4) == b"".join(os.fsencode(n) + b"\x00" for n in want[2:4]) and lst.names_blob(3, 3) == b""
    for broken in (blob[:-1], blob + b"zzz", b"b\x00a\x00", b"a\x00a\x00", b"\x00"):
        with pytest.raises(ValueError):
            sharding.DirListing.from_names_blob(str(d), broken, broken.count(b"\x00"))
    with pytest.raises(ValueError):
        sharding.DirListing.from_names_blob(str(d), blob, len(lst) + 1)
    assert len(sharding.DirListing.from_names_blob(str(d), b"", 0)) == 0
    lst.close(); lst.close()


def test_native_directory_listing_property_based(tmp_path):
    """Random name sets (shared prefixes, digits, mixed case, UTF-8, lengths 1..40): the library's order is sorted()'s."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from catfish_amd import sharding
    alphabet = st.sampled_from(list("0123456789abcXYZ_-.~") + ["é", "中", "\U0001f600"])
    name = st.builds(lambda p, t: p + t, st.sampled_from(["", "r", "read_", "read_0000", "channel_100_read_"]),
                     st.text(alphabet, min_size=1, max_size=24)).filter(lambda n: n not in (".", "..") and len(n.encode()) < 200)
    counter = [0]

    @settings(max_examples=60, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(st.sets(name, min_size=0, max_size=60))
    def check(names):
        counter[0] += 1
        d = tmp_path / ("d%d" % counter[0])
        d.mkdir()
        for n in names:
            (d / n).write_bytes(b"")
        lst = sharding.DirListing(str(d))
        assert lst.names() == sorted(names) == sorted(os.listdir(d))
        lst.close()
    check()


def test_bench_line_describes_its_ranks_and_refuses_to_call_a_rehearsal_a_measurement():
    """bench.describe_ranks on made-up rank records -- four ranks on four cards: a measurement (value kept,
    n_gpus 4); four ranks on ONE card: ``rehearsal: true``, ``value: null``, ``n_gpus: 1``, ``n_ranks: 4``, the legs marked, the
    whole-node rate withheld; overlapping CPU sets are seen; one process is one device and no rehearsal."""
    import bench

    def rank(r, uuid, cpus, host="node0"):
        return {"rank": r, "local_rank": r, "device_index": r, "pci_bus_id": "0000:%02x:00.0" % (10 + r), "uuid": uuid, "host": host,
                "cpus": cpus, "dt_s": 0.1, "ms_per_step": 3.2}

    def line(world):
        return {"value": 1.0e9, "n_gpus": world, "config": {"workload": "w"}, "host_to_host_pipeline": {"value": 3.0e8},
                "sharded_gather": {"value": 2.9e8, "n_gpus": world}, "cli_end_to_end": {"value": 2.8e8, "n_gpus": world}}

    real = bench.describe_ranks(line(4), [rank(r, "uuid%d" % r, "%d-%d" % (16 * r, 16 * r + 15)) for r in range(4)], 4, "nccl", None, 4, 8)
    assert real["rehearsal"] is False and real["value"] == 1.0e9 and real["n_gpus"] == 4 and real["distinct_devices"] == 4
    assert real["cpu_sets_disjoint"] is True and real["whole_node_end_to_end"] == 2.8e8 and real["host_to_host_value"] == 3.0e8
    assert real["collective"] == dict(real["collective"], backend="nccl", nccl_init_error=None, ranks_seen=4, ranks_seen_is_world=True)
    assert "4 GPU(s), one rank each" in real["config"]["parallelism"] and len(real["ranks"]) == 4

    same = bench.describe_ranks(line(4), [dict(rank(r, "uuid0", "%d-%d" % (16 * r, 16 * r + 15)), pci_bus_id="0000:0a:00.0") for r in range(4)], 4, "gloo",
                                "DistBackendError: Duplicate GPU detected", 4, 1)
    assert same["rehearsal"] is True and same["value"] is None and same["rehearsal_value"] == 1.0e9
    assert same["n_gpus"] == 1 and same["n_ranks"] == 4 and same["distinct_devices"] == 1
    assert same["whole_node_end_to_end"] is None and same["rehearsal_whole_node_end_to_end"] == 2.8e8
    assert "4 ranks on 1 device(s): REHEARSAL" in same["config"]["parallelism"]
    assert same["cli_end_to_end"]["rehearsal"] is True and same["cli_end_to_end"]["n_gpus"] == 1 and same["sharded_gather"]["n_ranks"] == 4
    assert same["collective"]["backend"] == "gloo" and "Duplicate GPU" in same["collective"]["nccl_init_error"]

    # two ranks on two hosts with equal uuids are two cards; overlapping CPU sets are reported; a failed parity gate keeps value null
    two = line(2)
    two.update(value=None, unverified_value=7.0)
    out = bench.describe_ranks(two, [dict(rank(0, "u", "0-7", "a"), pci_bus_id="0000:0a:00.0"), dict(rank(1, "u", "4-11", "b"), pci_bus_id="0000:0a:00.0")],
                               2, "gloo", None, 1, 1)
    assert out["rehearsal"] is False and out["distinct_devices"] == 2 and out["cpu_sets_disjoint"] is False and out["value"] is None
    assert out["collective"]["ranks_seen_is_world"] is False
    # a runtime that reports the same (or an empty) UUID for every card: the PCI addresses still tell the cards apart
    blank = bench.describe_ranks(line(4), [rank(r, "", "%d-%d" % (16 * r, 16 * r + 15)) for r in range(4)], 4, "nccl", None, 4, 4)
    assert blank["rehearsal"] is False and blank["distinct_devices"] == 4 and blank["value"] == 1.0e9
    one = bench.describe_ranks(line(1), [rank(0, "u", "0-63,128-191")], 1, None, None, 1, 1)
    assert one["rehearsal"] is False and one["n_gpus"] == 1 and one["value"] == 1.0e9 and "single process" in one["collective"]["what"]


def test_torch_operator_is_registered_and_has_no_cpu_path():
    """``torch.ops.catfish.resnetrnn_forward(x, packed_weights) -> Tensor`` exists with that schema; the packed tensor is
    the checkpoint's 74 inference tensors behind an 8-value header and unpacks to the same arrays; shape inference works without a
    device (meta / fake tensors); a CPU input is an error, never a slow answer."""
    import torch
    import catfish_amd.torch_ops as ops
    with np.load(os.path.join(ROOT, "tests", "golden", "ckpnt-30000-inference.npz")) as z:
        w = {k: z[k] for k in z.files}
    assert str(torch.ops.catfish.resnetrnn_forward.default._schema) == "catfish::resnetrnn_forward(Tensor x, Tensor packed_weights) -> Tensor"
    packed = ops.pack_weights(w)
    assert packed.dtype == torch.float32 and packed.shape == (8 + 197185,) and not packed.is_cuda          # 197 185 parameters
    assert sorted(ops.tensor_names()) == sorted(w) and len(ops.tensor_names()) == 74
    back, geom = ops.unpack_weights(packed)
    assert geom == dict(n_layers=3, layer_size=64, n_layers_res=2, layer_size_res=32)
    assert all(np.array_equal(back[k], w[k]) and back[k].shape == w[k].shape for k in w)
    assert torch.ops.catfish.resnetrnn_forward(torch.empty(10, 35, device="meta"), packed).shape == (350,)
    with pytest.raises(ValueError, match="MI355X only"):
        torch.ops.catfish.resnetrnn_forward(torch.zeros(4, 35), packed)
    bad = dict(w)
    del bad["conv1d_3/bias"]
    with pytest.raises(ValueError, match="conv1d_3/bias"):
        ops.pack_weights(bad)
    with pytest.raises(ValueError, match="shape"):
        ops.pack_weights(w, layer_size=32)
    for broken in (packed[:-1], packed.double(), torch.zeros(10), packed.reshape(1, -1)):
        with pytest.raises(ValueError):
            ops.unpack_weights(broken)
    from oracle import catfish_oracle as oracle
    rnn = oracle.random_weights(seed=1, n_layers=2, n_layers_res=0)                                          # the plain RNN type: no conv stack
    p2 = ops.pack_weights(rnn, n_layers=2, n_layers_res=0)
    b2, g2 = ops.unpack_weights(p2)
    assert g2["n_layers_res"] == 0 and sorted(b2) == sorted(ops.tensor_names(2, 0)) and b2[ops.tensor_names(2, 0)[0]].shape == (1 + 64, 128)


def test_torch_operator_keys_its_engines_on_the_weights_content(monkeypatch):
    """the engine cache was keyed on (storage address, version, numel, device); the allocator hands a freed packed
    tensor's address to the next one of the same size, so a second checkpoint of the same geometry silently ran with the first
    one's engine.  The key is a digest of the packed bytes now: pack A, use, delete, pack B into the SAME memory -> B's engine; equal
    content under another address -> the same engine; an in-place change of a live tensor -> a new engine."""
    import gc
    import torch
    import catfish_amd.engine as engine_mod
    import catfish_amd.torch_ops as ops
    from oracle import catfish_oracle as oracle
    built = []

    class FakeEngine(object):
        def __init__(self, weights, device=0, **geom):
            self.mark = float(weights["final_fully_connected/bias"][0])
            built.append(self.mark)

        def close(self):
            pass

    monkeypatch.setattr(engine_mod, "HipEngine", FakeEngine)
    ops.clear_engine_cache()
    wa, wb = oracle.random_weights(seed=11), oracle.random_weights(seed=12)
    wa["final_fully_connected/bias"] = np.array([0.25], np.float32)
    wb["final_fully_connected/bias"] = np.array([0.75], np.float32)
    reused = 0
    for _ in range(10):
        a = ops.pack_weights(wa)
        addr = a.untyped_storage().data_ptr()
        assert ops._engine_for(a, 0).mark == 0.25 and ops._engine_for(a, 0).mark == 0.25
        del a
        gc.collect()
        b = ops.pack_weights(wb)                                      # same size, version 0, very likely the same address
        reused += b.untyped_storage().data_ptr() == addr
        assert ops._engine_for(b, 0).mark == 0.75
        del b
        gc.collect()
    assert built == [0.25, 0.75]                                     # one engine per CONTENT, however often it was re-packed
    a = ops.pack_weights(wa)
    twin = a.clone()
    assert ops._engine_for(twin, 0) is ops._engine_for(a, 0)          # another address, equal bytes
    assert ops._engine_for(a, 1) is not ops._engine_for(a, 0)         # another device: its own engine
    a[-1] = 0.5                                                      # in place on a live tensor: torch's version counter moves
    assert ops._engine_for(a, 0).mark == 0.5 and ops._engine_for(twin, 0).mark == 0.25
    assert not ops._SEEN or all(ref() is not None for ref, _v, _k in ops._SEEN.values())      # dead tensors left the identity table
    for broken in (a.double(), a.reshape(1, -1), "not a tensor"):                # refused before anything is hashed or built
        with pytest.raises(ValueError, match="flat float32 CPU tensor"):
            ops._engine_for(broken, 0)
    ops.clear_engine_cache()
    assert not ops._ENGINES and not ops._SEEN


def test_file_batches_ramp_up_and_cover_every_read_once():
    """sharding._batches_by_samples: consecutive batches under the sample cap; with ``ramp`` the first batches are capped lower (the
    device starts sooner on a shard that begins with reading files) and every read still appears exactly once, in order."""
    from catfish_amd import sharding
    lens = [4096] * 3000
    plain = list(sharding._batches_by_samples(list(range(3000)), lens, 1120 * 4096))
    assert [len(b) for b in plain] == [1120, 1120, 760]
    ramped = list(sharding._batches_by_samples(list(range(3000)), lens, 1120 * 4096, ramp=sharding.RAMP))
    assert [len(b) for b in ramped] == [140, 420, 1120, 1120, 200] and sum(ramped, []) == list(range(3000))
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 5000, size=400).tolist()
    for ramp in ((), (0.125, 0.375), (0.001,)):
        got = list(sharding._batches_by_samples(list(range(400)), lens, 40000, ramp=ramp))
        assert sum(got, []) == list(range(400))
        assert all(len(b) == 1 or sum(lens[i] for i in b) <= 40000 for b in got)
    assert list(sharding._batches_by_samples([], [], 10, ramp=(0.5,))) == []
    assert list(sharding._batches_by_samples([0], [99], 10, ramp=(0.5,))) == [[0]]          # a read longer than the cap still gets a batch


def test_native_loader_reads_a_listing_block_without_path_strings(tmp_path):
    """cf_listing_load_npy_int16 (``DirListing`` entries [lo, hi) opened relative to the listing's directory) against ``infer.load_dac``
    on the same files: same samples, same lengths, in listing order; a block holding an entry that is not named ``*.npy`` or is not an
    int16 vector is refused with the entry named -- the caller then takes the general loader for that batch."""
    import ctypes as C
    from catfish_amd import _native as N, infer, sharding
    d = tmp_path / "reads"
    d.mkdir()
    rng = np.random.default_rng(2)
    want = {}
    for i, n in enumerate([4096, 1, 0, 35, 700, 12345, 2, 999]):
        name = "read_%03d.npy" % (7 * i % 8)
        np.save(d / name, rng.integers(-3000, 3000, size=n).astype(np.int16))
        want[name] = infer.load_dac(str(d / name))
    lst = sharding.DirListing(str(d))
    names = lst.names()
    assert names == sorted(want)
    lib = N.lib()

    def load(lo, hi, cap, threads=3):
        out = np.full(max(cap, 1), -7, dtype=np.int16)
        lengths = np.full(max(hi - lo, 1), -1, dtype=np.int64)
        total = C.c_int64(-1)
        rc = lib.cf_listing_load_npy_int16(lst._handle, lo, hi, out.ctypes.data_as(C.c_void_p), cap, lengths.ctypes.data_as(C.c_void_p),
                                           C.byref(total), threads)
        return rc, out, lengths[:hi - lo], total.value

    everything = np.concatenate([want[n] for n in names])
    rc, out, lengths, total = load(0, len(names), len(everything))
    assert rc == 0 and total == len(everything) and lengths.tolist() == [len(want[n]) for n in names]
    assert np.array_equal(out[:total], everything)
    rc, out, lengths, total = load(2, 5, 20000, threads=1)
    assert rc == 0 and np.array_equal(out[:total], np.concatenate([want[n] for n in names[2:5]]))
    assert load(3, 3, 0)[0] == 0 and load(3, 3, 0)[3] == 0
    rc, _out, _l, total = load(0, len(names), len(everything) - 1)                      # does not fit: refused, the need reported
    assert rc == N.CF_ERR_INVALID and total == len(everything)
    assert load(0, len(names) + 1, 10)[0] == N.CF_ERR_INVALID                            # range off the end
    np.save(d / "read_100.npy", np.zeros((2, 3), np.int16))                              # not a vector
    (d / "notes.txt").write_text("hello")                                                # not a .npy
    lst2 = sharding.DirListing(str(d))
    n2 = lst2.names()
    total = C.c_int64(0)
    for bad in ("read_100.npy", "notes.txt"):
        k = n2.index(bad)
        out = np.zeros(64, np.int16)
        lengths = np.zeros(1, np.int64)
        rc = lib.cf_listing_load_npy_int16(lst2._handle, k, k + 1, out.ctypes.data_as(C.c_void_p), 64, lengths.ctypes.data_as(C.c_void_p),
                                           C.byref(total), 2)
        assert rc == N.CF_ERR_INVALID and bad.encode() in lib.cf_last_error()
