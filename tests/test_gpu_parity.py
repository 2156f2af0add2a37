"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerance: per-sample sigmoid outputs within 1e-4 of the fp64 oracle
(BASELINE.json north_star); in practice fp32 MFMA lands ~1e-6.
"""
import numpy as np
import pytest

from oracle import catfish_oracle as oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def engine(ckpt_weights):
    from catfish_amd.engine import HipEngine
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096)
    yield eng
    eng.close()


def test_golden_read_stages(engine, golden_read):
    """Stage-by-stage parity on the first tile of the golden read (real checkpoint)."""
    x = golden_read["x"]
    probs = engine.infer_host(x)
    for stage, key in ((0, "res0_w16"), (1, "res1_w16"), (2, "gru0_w16"), (3, "gru1_w16")):
        got = engine.debug_stage(stage, 16)
        err = np.abs(got - golden_read[key]).max()
        assert err < 2e-5, "stage %s max err %g" % (key, err)
    err = np.abs(probs.astype(np.float64) - golden_read["probs_fp64"]).max()
    assert err < TOL, err


def test_golden_read_probs(engine, golden_read):
    probs = engine.infer_host(golden_read["x"])
    assert probs.shape == (118 * 35,)
    err = np.abs(probs.astype(np.float64) - golden_read["probs_fp64"]).max()
    assert err < TOL, err
    # label match rate vs the fp32 oracle
    assert np.mean((probs >= 0.5) == (golden_read["probs_fp32"] >= 0.5)) == 1.0


def test_hip_path_matches_reference_graph(engine, graph_golden):
    """The HIP path against the outputs of the reference's own TF graph (ckpnt-30000.meta interpreted in
    numpy, tests/golden/make_graph_golden.py) -- no oracle in between."""
    probs = engine.infer_host(graph_golden["x"])
    assert np.abs(probs.astype(np.float64) - graph_golden["ckpt_f64_probs"]).max() < TOL
    assert np.abs(probs - graph_golden["ckpt_f32_probs"]).max() < 2e-5
    for stage, key in ((0, "res0"), (1, "res1"), (2, "gru0"), (3, "gru1")):
        ref = graph_golden["ckpt_f64_" + key]
        got = engine.debug_stage(stage, ref.shape[0])
        assert got.shape == ref.shape, key
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-5), "stage %s max err %g" % (key, np.abs(got - ref).max())


@pytest.mark.parametrize("n_windows", [1, 15, 16, 17, 127, 128, 129, 1000])
def test_ragged_window_counts_random_weights(n_windows):
    """Random weights (non-trivial BN stats/biases) and ragged tile/workgroup tails."""
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=n_windows)
    rng = np.random.default_rng(n_windows)
    x = rng.normal(0, 1.5, size=(n_windows, 35)).astype(np.float32)
    eng = HipEngine(w, device=0, max_windows_per_pass=512)   # 1000 windows -> two passes
    try:
        got = eng.infer_host(x)
    finally:
        eng.close()
    want = oracle.forward(x, w, np.float64)
    err = np.abs(got - want).max()
    assert err < TOL, err


def _regime_sizes():
    """Window counts either side of every launch-regime switch of the fp32 path, read from the library
    (cf_launch_regimes) instead of hard-coded: hoisted x projection up to 3*CUs/16 tiles (768 windows on 256 CUs),
    cooperative latency-mode kernels up to CUs tiles (4096 windows), one wave per tile beyond."""
    from catfish_amd.engine import HipEngine
    eng = HipEngine(oracle.random_weights(seed=5), device=0, max_windows_per_pass=8192)
    try:
        return eng.launch_regimes()
    finally:
        eng.close()


@pytest.mark.parametrize("which", ["hoist_max", "hoist_max+16", "coop_max", "coop_max+16", "coop_max+3"])
def test_launch_regime_boundaries(which):
    """One pass each side of the two regime switches of the fp32 path, plus a ragged last tile in throughput mode
    (coop_max + 3 windows), against the fp64 oracle at the 1e-4 gate and the fp32 oracle at 2e-5."""
    from catfish_amd.engine import HipEngine
    reg = _regime_sizes()
    assert reg["hoist_max"] == 16 * (3 * reg["n_cu"] // 16) and reg["coop_max"] == 16 * reg["n_cu"]
    name, _, extra = which.partition("+")
    n_windows = reg[name] + int(extra or 0)
    w = oracle.random_weights(seed=5)
    rng = np.random.default_rng(n_windows)
    x = rng.normal(0, 1.5, size=(n_windows, 35)).astype(np.float32)
    eng = HipEngine(w, device=0, max_windows_per_pass=8192)
    try:
        got = eng.infer_host(x)
    finally:
        eng.close()
    assert np.abs(got - oracle.forward(x, w, np.float64)).max() < TOL
    assert np.abs(got - oracle.forward(x, w, np.float32)).max() < 2e-5


@pytest.mark.parametrize("n_windows", [30208 - 7, 30208, 30208 + 7])
def test_throughput_kernel_at_benchmark_size(ckpt_weights, n_windows):
    """The headline kernel (one wave per tile, 3.7 rounds on 256 CUs) at BASELINE configs[1]'s launch size and with
    ragged last tiles either side of it: the last 48 windows (the ragged tile and its neighbours) plus 80 random
    ones against the fp64 oracle; all windows bit-identical to the same windows inferred in a small call."""
    from catfish_amd.engine import HipEngine
    rng = np.random.default_rng(n_windows)
    x = rng.normal(0, 1.5, size=(n_windows, 35)).astype(np.float32)
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=32768)
    try:
        got = eng.infer_host(x).reshape(n_windows, 35)
        # throughput mode runs residual blocks 0 and 1 as ONE launch: block 1's output is checked against the oracle,
        # block 0's is never stored
        res1 = eng.debug_stage(1, 48)
        with pytest.raises(ValueError, match="not materialised"):
            eng.debug_stage(0, 48)
        idx = np.concatenate([np.arange(n_windows - 48, n_windows), rng.integers(0, n_windows - 48, size=80)])
        small = eng.infer_host(x[idx]).reshape(len(idx), 35)        # 128 windows: latency-mode kernels (one launch per block)
        res1_small = eng.debug_stage(1, 128)
        eng.infer_host(x[:48])
        assert np.array_equal(eng.debug_stage(1, 48), res1)         # fused two-block launch == two launches, bit for bit
        assert eng.debug_stage(0, 48).shape == (48, 35, 32)
    finally:
        eng.close()
    want, stages = oracle.forward(x[idx], ckpt_weights, np.float64, return_stages=True)
    want = want.reshape(len(idx), 35)
    assert np.abs(res1_small - stages["res1"]).max() < 2e-5
    assert np.abs(got[idx] - want).max() < TOL
    assert np.array_equal(got[idx], small)                          # every regime gives the same bits
    assert np.isfinite(got).all()


def test_device_path_matches_host_path(engine, golden_read):
    torch = pytest.importorskip("torch")
    x = torch.from_numpy(golden_read["x"]).cuda()
    out = engine.infer_device(x)
    torch.cuda.synchronize()
    host = engine.infer_host(golden_read["x"])
    assert np.array_equal(out.cpu().numpy(), host)


def test_empty_and_bad_shapes(engine):
    assert engine.infer_host(np.zeros((0, 35), np.float32)).shape == (0,)
    with pytest.raises(ValueError):
        engine.infer_host(np.zeros((4, 34), np.float32))


@pytest.mark.parametrize("precision,tol,min_match", [("bf16x3", 1e-4, 0.9995), ("bf16", 3e-2, 0.99)])
def test_bf16_modes_on_golden_read(ckpt_weights, golden_read, precision, tol, min_match):
    """bf16x3 (split-operand fp32 emulation) keeps the 1e-4 gate; plain bf16 (config 4) is judged by
    label match rate and a loose probability bound."""
    from catfish_amd.engine import HipEngine
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, precision=precision)
    try:
        probs = eng.infer_host(golden_read["x"])
    finally:
        eng.close()
    err = np.abs(probs.astype(np.float64) - golden_read["probs_fp64"]).max()
    match = np.mean((probs >= 0.5) == (golden_read["probs_fp32"] >= 0.5))
    print(precision, "max|dp|", err, "label match", match)
    assert err < tol, err
    assert match >= min_match, match


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-4), ("bf16", 5e-2)])
@pytest.mark.parametrize("n_windows", [1, 31, 32, 33, 255, 257, 1000])
def test_bf16_modes_ragged_random_weights(precision, tol, n_windows):
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=100 + n_windows)
    rng = np.random.default_rng(n_windows)
    x = rng.normal(0, 1.5, size=(n_windows, 35)).astype(np.float32)
    eng = HipEngine(w, device=0, max_windows_per_pass=512, precision=precision)
    try:
        got = eng.infer_host(x)
    finally:
        eng.close()
    want = oracle.forward(x, w, np.float64)
    err = np.abs(got - want).max()
    assert np.isfinite(got).all()
    assert err < tol, err


@pytest.mark.parametrize("n_windows", [16, 129, 2000])
def test_fused_single_launch_matches_per_layer_launches(ckpt_weights, n_windows):
    """fuse_layers=True: all biGRU layers in one launch (agent-scope flags + dynamic queues) is bit-identical
    to one launch per layer; repeated to catch a stale hand-off."""
    from catfish_amd.engine import HipEngine
    rng = np.random.default_rng(n_windows)
    x = rng.normal(0, 1.5, size=(n_windows, 35)).astype(np.float32)
    a = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, fuse_layers=False)
    b = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, fuse_layers=True)
    try:
        ref = a.infer_host(x)
        for _ in range(5):
            assert np.array_equal(b.infer_host(x), ref)
        assert np.array_equal(b.debug_stage(3, min(n_windows, 64)), a.debug_stage(3, min(n_windows, 64)))
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_fused_bf16_residual_stack_is_bit_identical_to_two_launches(ckpt_weights, precision, monkeypatch):
    """res_stack2_bf16_kernel (blocks 0 and 1 in one launch, block 0's bf16-rounded output handed over in registers, a tile's
    positions cut into chunks of one wave each) against res_block_bf16_kernel<true> + <false> (CATFISH_RES_FUSE=0 behind the
    debug switch): the same bits at every call size -- one chunk per position (a single read), a few positions per chunk,
    the benchmark's launch +- a ragged tile, and a pass big enough for one chunk per tile; then a deeper stack (3 blocks:
    the third one still runs on the per-block kernel) against the fp64 oracle."""
    from catfish_amd.engine import HipEngine
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=140000, precision=precision)
    try:
        for n in (1, 31, 118, 1000, 4097, 30208 + 7, 131072 + 33):
            x = np.random.default_rng(n).normal(0, 1.4, size=(n, 35)).astype(np.float32)
            monkeypatch.delenv("CATFISH_RES_FUSE", raising=False)
            fused, fused_logits = eng.infer_host(x, return_logits=True)
            monkeypatch.setenv("CATFISH_RES_FUSE", "0")
            two, two_logits = eng.infer_host(x, return_logits=True)
            assert np.array_equal(fused, two) and np.array_equal(fused_logits, two_logits), n
            assert np.isfinite(fused).all()
            if precision == "bf16" and n in (31, 4097):                         # the two-tiles-per-wave variant (debug knob)
                monkeypatch.delenv("CATFISH_RES_FUSE")
                monkeypatch.setenv("CATFISH_RES_TPW", "2")
                assert np.array_equal(eng.infer_host(x), two), n
                monkeypatch.delenv("CATFISH_RES_TPW")
        monkeypatch.delenv("CATFISH_RES_FUSE", raising=False)
        x = np.random.default_rng(5).normal(0, 1.4, size=(200, 35)).astype(np.float32)
        want = oracle.forward(x, ckpt_weights, np.float64)
        assert np.abs(eng.infer_host(x) - want).max() < (3e-2 if precision == "bf16" else 1e-4)
    finally:
        eng.close()
    w = oracle.random_weights(seed=31, n_layers_res=3)
    eng = HipEngine(w, n_layers_res=3, device=0, max_windows_per_pass=4096, precision=precision)
    try:
        x = np.random.default_rng(6).normal(0, 1.2, size=(333, 35)).astype(np.float32)
        want = oracle.forward(x, w, np.float64, n_layers_res=3)
        assert np.abs(eng.infer_host(x) - want).max() < (3e-2 if precision == "bf16" else 1e-4)
    finally:
        eng.close()


@pytest.mark.parametrize("h,c,n_layers,n_layers_res,n", [
    (16, 16, 1, 1, 70), (32, 64, 2, 1, 333), (128, 16, 2, 2, 90), (256, 128, 1, 1, 40), (64, 256, 1, 2, 50),
    (48, 80, 3, 3, 100), (80, 48, 2, 1, 77), (96, 16, 1, 1, 30), (32, 0, 2, 0, 200),
    (64, 128, 3, 1, 100), (64, 16, 2, 2, 300), (64, 64, 3, 1, 5000), (64, 256, 2, 2, 4100), (128, 0, 1, 0, 60), (64, 32, 5, 4, 64), (16, 32, 2, 2, 4500)])
def test_any_size_models_match_oracle(h, c, n_layers, n_layers_res, n):
    """The model classes accept any layer_size / layer_size_res (the reference's hyper-parameter search draws 16..256 and
    1..5 / 1..11 layers, networks/train_validate.py:66-111); everything but the shipped 64 / 32 geometry runs on the
    any-size kernels (csrc/generic.hpp) -- except that 64-unit layers with 16 / 32 / 128 inputs inside such a model go to the
    LDS-resident kernel, in whichever launch regime the call size selects (the 64-unit rows).  Random glorot weights, ragged
    window counts, the plain RNN type (c = 0) included: probabilities within 1e-4 of the fp64 oracle, logits consistent,
    batch-split invariant."""
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=100 + h + c, layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res)
    rng = np.random.default_rng(h * 7 + c)
    x = rng.normal(0, 1.3, size=(n, 35)).astype(np.float32)
    eng = HipEngine(w, layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res, device=0,
                    max_windows_per_pass=2048)
    try:
        assert eng.launch_regimes()["coop_max"] == 0 or (h == 64 and c in (0, 32))     # shipped geometry = tuned kernels
        got, logits = eng.infer_host(x, return_logits=True)
        m = min(n, 160)
        want = oracle.forward(x[:m], w, np.float64, n_layers=n_layers, n_layers_res=n_layers_res)
        assert got.shape == (n * 35,) and np.isfinite(got).all()
        assert np.abs(got[:m * 35] - want).max() < TOL
        assert np.abs(1.0 / (1.0 + np.exp(-logits.astype(np.float64))) - got).max() < 1e-6
        if n > m:                       # the tail of a multi-pass call (2048-window passes) against the oracle too
            want_t = oracle.forward(x[n - 40:], w, np.float64, n_layers=n_layers, n_layers_res=n_layers_res)
            assert np.abs(got[(n - 40) * 35:] - want_t).max() < TOL
        part = eng.infer_host(x[3:n // 2 + 1])
        assert np.array_equal(part, got[3 * 35:(n // 2 + 1) * 35])
        eng.check_error()
    finally:
        eng.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_random_call_sizes_and_offsets_are_bit_identical(ckpt_weights, precision):
    """Differential fuzz over the launch regimes (hoisted / cooperative / throughput / multi-pass): 80 seeded calls of
    log-uniform size 1..40 000 windows at random offsets into one fixed input must reproduce, bit for bit, the
    corresponding slice of a single 40 000-window call -- a window's result may not depend on the call it travels in."""
    torch = pytest.importorskip("torch")
    from catfish_amd.engine import HipEngine
    n_all = 40000
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=32768, precision=precision)
    try:
        g = torch.Generator(device="cpu").manual_seed(11)
        x = torch.randn(n_all, 35, generator=g).mul_(1.4).cuda()
        full = eng.infer_device(x).view(n_all, 35).clone()
        rng = np.random.default_rng(12)
        sizes = np.unique(np.concatenate([np.exp(rng.uniform(0, np.log(n_all), size=72)).astype(int), [1, 15, 16, 17, 32768, 32769, n_all, 4097]]))
        for n in sizes:
            n = int(min(max(n, 1), n_all))
            o = int(rng.integers(0, n_all - n + 1))
            got = eng.infer_device(x[o:o + n].contiguous()).view(n, 35)
            assert torch.equal(got, full[o:o + n]), (precision, n, o)
        eng.check_error()
    finally:
        eng.close()


def test_unsupported_geometries_are_refused():
    """Sizes the any-size kernels cannot tile (not a multiple of 16, above 256) and bf16 for a non-shipped geometry raise
    ValueError with the library's message; nothing is allocated or left behind."""
    from catfish_amd.engine import HipEngine
    for kw in (dict(layer_size=24, layer_size_res=32), dict(layer_size=64, layer_size_res=40), dict(layer_size=272, layer_size_res=32),
               dict(layer_size=8, layer_size_res=16)):
        w = oracle.random_weights(seed=1, n_layers=1, n_layers_res=1, **kw)
        with pytest.raises(ValueError, match="multiple of 16"):
            HipEngine(w, n_layers=1, n_layers_res=1, device=0, max_windows_per_pass=64, **kw)
    w = oracle.random_weights(seed=1, layer_size=128, layer_size_res=32, n_layers=1, n_layers_res=1)
    with pytest.raises(ValueError, match="bf16"):
        HipEngine(w, layer_size=128, layer_size_res=32, n_layers=1, n_layers_res=1, device=0, max_windows_per_pass=64, precision="bf16")


@pytest.mark.parametrize("h,c,n_layers,n_layers_res,n", [(128, 64, 2, 1, 9001), (64, 64, 2, 1, 8400), (128, 0, 2, 0, 8200)])
def test_any_size_two_tile_kernel_is_bit_identical_to_the_one_tile_kernel(h, c, n_layers, n_layers_res, n, monkeypatch):
    """Big launches of 64- and 128-unit models run two 16-window tiles per wave (gen_gru2_kernel: every weight fragment feeds
    eight MFMAs); the result must equal the one-tile kernel's bit for bit (CATFISH_GEN_ONE_TILE=1 and, independently, small
    calls, which never use it), and a sample is checked against the fp64 oracle."""
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=60 + h, layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res)
    x = np.random.default_rng(h).normal(0, 1.3, size=(n, 35)).astype(np.float32)
    kw = dict(layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res, device=0, max_windows_per_pass=16384)
    two = HipEngine(w, **kw)
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")          # the library reads its A/B knobs only behind this switch
    monkeypatch.setenv("CATFISH_GEN_ONE_TILE", "1")
    one = HipEngine(w, **kw)
    monkeypatch.delenv("CATFISH_GEN_ONE_TILE")
    try:
        a, b = two.infer_host(x), one.infer_host(x)
        assert np.array_equal(a, b)
        assert np.array_equal(two.infer_host(x[5000:5300]), a[5000 * 35:5300 * 35])       # a small call: the one-tile kernel
        idx = np.concatenate([np.arange(0, 40), np.arange(n - 40, n), [4095, 4096, 8191]])
        idx = idx[idx < n]
        want = oracle.forward(x[idx], w, np.float64, n_layers=n_layers, n_layers_res=n_layers_res).reshape(len(idx), 35)
        assert np.abs(a.reshape(n, 35)[idx] - want).max() < TOL
        two.check_error()
    finally:
        two.close(); one.close()


def test_any_size_path_agrees_with_the_tuned_kernels_on_the_checkpoint(ckpt_weights, monkeypatch):
    """CATFISH_GENERIC=1 sends the shipped geometry through the any-size kernels too: same checkpoint, same reads, the two
    implementations agree to 2e-6 and both sit within 1e-4 of the fp64 oracle; bf16 is refused on that path."""
    from catfish_amd.engine import HipEngine
    x = np.random.default_rng(21).normal(0, 1.5, size=(354, 35)).astype(np.float32)
    tuned = HipEngine(ckpt_weights, device=0, max_windows_per_pass=1024)
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")          # the library reads its A/B knobs only behind this switch
    monkeypatch.setenv("CATFISH_GENERIC", "1")
    try:
        generic = HipEngine(ckpt_weights, device=0, max_windows_per_pass=1024)
        with pytest.raises(ValueError):
            HipEngine(ckpt_weights, device=0, max_windows_per_pass=1024, precision="bf16")
    finally:
        monkeypatch.delenv("CATFISH_GENERIC")
    try:
        assert generic.launch_regimes()["coop_max"] == 0 and tuned.launch_regimes()["coop_max"] > 0
        a, b = tuned.infer_host(x), generic.infer_host(x)
        want = oracle.forward(x[:118], ckpt_weights, np.float64)
        assert np.abs(a - b).max() < 2e-6
        assert np.abs(b[:118 * 35] - want).max() < TOL
    finally:
        tuned.close(); generic.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_million_window_launch_index_arithmetic(ckpt_weights, precision):
    """One launch of 1 920 021 windows (67 M samples, 86 GB of workspace out of the card's 288 GB): the activation buffers
    cross 2^31 bytes (tile 7 490), 2^31 floats (tile 29 960) and 2^31 16-byte fragments (tile 119 837).  Windows either
    side of each boundary, the ragged last tile and a random sample must equal the same windows run as a small launch,
    bit for bit (a window's arithmetic does not depend on where it sits), and a few are checked against the fp64 oracle."""
    torch = pytest.importorskip("torch")
    from catfish_amd.engine import HipEngine
    n = 119837 * 16 + 2629                                     # 1 920 021
    free, _ = torch.cuda.mem_get_info(0)
    if free < 100 * 2 ** 30:
        pytest.skip("needs 100 GB of free device memory")
    big = HipEngine(ckpt_weights, device=0, max_windows_per_pass=n, precision=precision)
    small = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, precision=precision)
    try:
        g = torch.Generator(device="cuda").manual_seed(9)
        x = torch.randn(n, 35, device="cuda", generator=g).mul_(1.5)
        got = big.infer_device(x).view(n, 35)
        big.check_error()
        assert torch.isfinite(got).all()
        cuts = [0, 7490 * 16, 29960 * 16, 119837 * 16, n // 2 + 5]
        for c in cuts:
            lo, hi = max(0, c - 77), min(n, c + 83)
            assert torch.equal(small.infer_device(x[lo:hi].contiguous()).view(hi - lo, 35), got[lo:hi]), (precision, c)
        assert torch.equal(small.infer_device(x[n - 300:].contiguous()).view(300, 35), got[n - 300:])
        idx = torch.randint(0, n, (2048,), device="cuda", generator=g)
        assert torch.equal(small.infer_device(x[idx].contiguous()).view(2048, 35), got[idx])
        small.check_error()
        pick = torch.tensor([0, 7490 * 16 + 1, 29960 * 16 - 1, 119837 * 16, 119837 * 16 + 17, n - 1], device="cuda")
        want = oracle.forward(x[pick].cpu().numpy(), ckpt_weights, np.float64).reshape(len(pick), 35)
        assert np.abs(got[pick].cpu().numpy() - want).max() < (TOL if precision == "fp32" else 3e-2)
    finally:
        big.close(); small.close()


@pytest.mark.timeout(900)
def test_fused_auto_regime_is_bit_identical_and_error_free(ckpt_weights):
    """The regime the CLI ships for big jobs (cli.py: 131 072-window launches, fuse_layers = auto): 768+ eight-tile
    groups per layer on 256 CUs, i.e. three layer pools of workgroups that cannot all be resident, layer l+1 waiting
    on agent-scope flags.  fuse_layers=None (auto) and =True must equal fuse_layers=False bit for bit, either side of
    the auto threshold (read from cf_launch_regimes) and at 131 072 windows with a ragged tail, repeated; the sticky
    device error flag must stay clear; a sample including the ragged tile is checked against the fp64 oracle."""
    from catfish_amd.engine import HipEngine
    cap = 131072
    per_layer = HipEngine(ckpt_weights, device=0, max_windows_per_pass=cap, fuse_layers=False)
    auto = HipEngine(ckpt_weights, device=0, max_windows_per_pass=cap, fuse_layers=None)
    forced = HipEngine(ckpt_weights, device=0, max_windows_per_pass=cap, fuse_layers=True)
    try:
        fmin = auto.launch_regimes()["fuse_auto_min"]
        assert fmin == (8 * 6 * (auto.launch_regimes()["n_cu"] // 2) - 8) * 16 + 1          # 98 177 windows on 256 CUs
        assert per_layer.launch_regimes()["fuse_auto_min"] == 0 and forced.launch_regimes()["fuse_auto_min"] == 1
        rng = np.random.default_rng(42)
        x_all = rng.normal(0, 1.5, size=(cap, 35)).astype(np.float32)
        for n in (fmin - 1, fmin, fmin + 127, cap - 5, cap):
            x = x_all[:n]
            ref = per_layer.infer_host(x)
            for rep in range(2):
                auto.profile_enable(True)
                auto.profile_reset()
                got = auto.infer_host(x)
                slots = auto.profile_read()
                auto.profile_enable(False)
                assert ("gru_fused" in slots) == (n >= fmin), (n, sorted(slots))      # the regime under test really ran
                assert np.array_equal(got, ref), (n, rep)
                auto.check_error()
            assert np.array_equal(forced.infer_host(x), ref), n
            forced.check_error()
        # oracle spot check of the fused result at the largest ragged size
        n = cap - 5
        got = auto.infer_host(x_all[:n]).reshape(n, 35)
        idx = np.concatenate([np.arange(n - 40, n), rng.integers(0, n - 40, size=56)])
        want = oracle.forward(x_all[idx], ckpt_weights, np.float64).reshape(len(idx), 35)
        assert np.abs(got[idx] - want).max() < TOL
    finally:
        per_layer.close(); auto.close(); forced.close()


def test_internal_streams_option_is_bit_identical(ckpt_weights):
    """cf_hparams.n_streams > 1 (sub-batches of one call on internal streams) gives the same bits."""
    from catfish_amd.engine import HipEngine
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1.5, size=(5000, 35)).astype(np.float32)
    a = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, n_streams=1)
    b = HipEngine(ckpt_weights, device=0, max_windows_per_pass=4096, n_streams=3)
    try:
        ref = a.infer_host(x)
        for _ in range(3):
            assert np.array_equal(b.infer_host(x), ref)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("n_layers,n_layers_res", [(1, 1), (2, 3), (4, 2), (1, 0), (2, 0)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_other_depths(n_layers, n_layers_res, precision):
    """Any number of biGRU layers / residual blocks (the reference's random search draws 1-6 and 1-12,
    networks/train_validate.py:67-111); widths stay 64 / 32."""
    from catfish_amd.engine import HipEngine
    if n_layers_res == 0 and precision != "fp32":
        pytest.skip("plain RNN type is fp32 only")
    w = oracle.random_weights(seed=7 * n_layers + n_layers_res, n_layers=n_layers, n_layers_res=n_layers_res)
    rng = np.random.default_rng(3)
    for n in (50, 3000, 4200):                 # latency mode with hoisted x projection, cooperative kernels, throughput kernels (> CUs tiles)
        x = rng.normal(0, 1.3, size=(n, 35)).astype(np.float32)
        eng = HipEngine(w, n_layers=n_layers, n_layers_res=n_layers_res, device=0, max_windows_per_pass=4096,
                        precision=precision)
        try:
            got = eng.infer_host(x)
        finally:
            eng.close()
        want = oracle.forward(x, w, np.float64, n_layers=n_layers, n_layers_res=n_layers_res)
        assert np.abs(got - want).max() < 1e-4, (n, np.abs(got - want).max())


def test_bf16x3_pipelined_kernel_agrees_with_unpipelined(ckpt_weights, monkeypatch):
    """gru_bf16x3_pipe_kernel (one wave per SIMD, every vector instruction placed in an MFMA gap by a compile-time schedule)
    against gru_layer_bf16_kernel<.,.,2> (CATFISH_BF16_PIPE=0 behind the debug switch).  Same products in the same order per
    accumulator, same activation arithmetic -- but not the same bits: the round-1 kernel leaves the lo part of r*h to the
    compiler, which contracts most (not all) of its `r*h - hi` into an fma, while the pipelined kernel pins every operation.
    That moves single bf16 lo parts by one ulp (2^-17 of the value).  The ISA of both is kept in
    profiles/r05_x3_contraction_isa.txt: 11 v_pk_fma_f32 with neg_lo/neg_hi on the addend + 2 v_fma_f32 ..., -v in the round-1
    kernel (24 of a lane's 32 r*h values per step), none in the pipelined one (v_mul_f32, v_cvt_pk_bf16_f32, v_sub_f32).  So: the two agree to 4e-6 in probability at every call
    size -- one ragged tile, a few tiles, fewer tiles than waves, the benchmark's launch +- a ragged tile, a multi-round
    launch (a lost or stale lo fragment anywhere shows as >= 1e-4) -- and each stays inside the 1e-4 gate of the fp64 oracle."""
    from catfish_amd.engine import HipEngine
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=70000, precision="bf16x3")
    try:
        for n in (1, 33, 118, 1000, 4097, 30208 + 7, 65536 + 31):
            x = np.random.default_rng(n).normal(0, 1.4, size=(n, 35)).astype(np.float32)
            monkeypatch.delenv("CATFISH_BF16_PIPE", raising=False)
            new, new_logits = eng.infer_host(x, return_logits=True)
            monkeypatch.setenv("CATFISH_BF16_PIPE", "0")
            old, old_logits = eng.infer_host(x, return_logits=True)
            assert np.isfinite(new).all() and np.isfinite(new_logits).all(), n
            assert np.abs(new - old).max() < 4e-6, (n, np.abs(new - old).max())
            m = min(n, 300)
            want = oracle.forward(x[-m:], ckpt_weights, np.float64)
            assert np.abs(new[-m * 35:] - want).max() < TOL and np.abs(old[-m * 35:] - want).max() < TOL, n
            assert np.abs(1.0 / (1.0 + np.exp(-new_logits.astype(np.float64))) - new).max() < 1e-6
    finally:
        eng.close()
    # another geometry of the stack around it: random weights, the oracle as the judge
    w = oracle.random_weights(seed=77)
    eng = HipEngine(w, device=0, max_windows_per_pass=4096, precision="bf16x3")
    try:
        x = np.random.default_rng(8).normal(0, 1.3, size=(777, 35)).astype(np.float32)
        assert np.abs(eng.infer_host(x) - oracle.forward(x, w, np.float64)).max() < TOL
    finally:
        eng.close()


def test_xproj_through_lds_is_bit_identical(ckpt_weights, monkeypatch):
    """Small calls hoist the x projection out of the recurrence (gru_coop.hpp).  gru_xproj_lds_kernel (weights staged in LDS,
    a chunk of steps per workgroup) against round 2's gru_xproj_kernel (every wave its fragments straight from L2;
    CATFISH_XPROJ_LDS=0 behind the debug switch): same bias-first, k-ascending accumulation, so the same bits -- at one tile,
    a single read, the chunk planner's other regimes and the last hoisted size; and the hoisted calls equal the same
    windows inside a throughput-regime call."""
    from catfish_amd.engine import HipEngine
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=8192)
    try:
        hoisted_max = eng.launch_regimes()["hoist_max"]
        x = np.random.default_rng(4).normal(0, 1.4, size=(8000, 35)).astype(np.float32)
        big = eng.infer_host(x)                                                  # throughput kernels
        for n in (1, 16, 118, 130, 300, 512, hoisted_max):
            monkeypatch.delenv("CATFISH_XPROJ_LDS", raising=False)
            new, new_logits = eng.infer_host(x[:n], return_logits=True)
            monkeypatch.setenv("CATFISH_XPROJ_LDS", "0")
            old, old_logits = eng.infer_host(x[:n], return_logits=True)
            assert np.array_equal(new, old) and np.array_equal(new_logits, old_logits), n
            assert np.array_equal(new, big[:n * 35]), n
    finally:
        eng.close()


@pytest.mark.parametrize("n_layers,n", [(1, 200), (2, 1100)])
def test_bf16x3_pipelined_kernel_at_other_depths(n_layers, n):
    """The pipelined bf16x3 kernel's other instantiations: a one-layer stack runs the 32-input kernel with the fused dense
    partial (<32, true>), two layers run <32, false> and <128, true> back to back; random weights against the fp64 oracle,
    and a call split in two gives the same bits as one."""
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=40 + n_layers, n_layers=n_layers)
    x = np.random.default_rng(n_layers).normal(0, 1.3, size=(n, 35)).astype(np.float32)
    eng = HipEngine(w, n_layers=n_layers, device=0, max_windows_per_pass=4096, precision="bf16x3")
    try:
        got = eng.infer_host(x)
        m = min(n, 200)
        assert np.abs(got[:m * 35] - oracle.forward(x[:m], w, np.float64, n_layers=n_layers)).max() < TOL
        assert np.array_equal(np.concatenate([eng.infer_host(x[:n // 3]), eng.infer_host(x[n // 3:])]), got)
    finally:
        eng.close()


def test_torch_operator_resnetrnn_forward(ckpt_weights, golden_read):
    """the replaced call (rnn_class.py:214-216) as ``torch.ops.catfish.resnetrnn_forward(x, packed_weights)`` -- device
    tensor in, device tensor out on PyTorch's current stream, the same C ABI underneath: bit-identical to ``HipEngine.infer_device``,
    within 1e-4 of the fp64 oracle; the engine is built once per packed tensor; another geometry packs and runs too; a side stream
    is honoured."""
    import torch
    import catfish_amd.torch_ops as ops
    from catfish_amd.engine import HipEngine
    ops.clear_engine_cache()
    packed = ops.pack_weights(ckpt_weights)
    x = torch.from_numpy(np.ascontiguousarray(golden_read["x"], dtype=np.float32)).cuda()              # [118, 35, 1]
    got = torch.ops.catfish.resnetrnn_forward(x, packed)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == (x.shape[0] * 35,)
    want = oracle.forward(golden_read["x"], ckpt_weights, np.float64)
    assert np.abs(got.cpu().numpy() - want).max() < TOL
    eng = HipEngine(ckpt_weights, device=0)
    try:
        assert torch.equal(got, eng.infer_device(x))
    finally:
        eng.close()
    assert len(ops._ENGINES) == 1
    again = torch.ops.catfish.resnetrnn_forward(x.reshape(-1, 35), packed)                              # [N, 35] is accepted as well
    assert torch.equal(again, got) and len(ops._ENGINES) == 1                                           # same tensor: same engine
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        on_side = torch.ops.catfish.resnetrnn_forward(x, packed)
    side.synchronize()
    assert torch.equal(on_side, got)
    # another geometry of the operator surface (one biGRU layer, one residual block, random weights)
    w = oracle.random_weights(seed=3, n_layers=1, n_layers_res=1)
    small = ops.pack_weights(w, n_layers=1, n_layers_res=1)
    xs = torch.randn(50, 35, device="cuda")
    ys = torch.ops.catfish.resnetrnn_forward(xs, small)
    assert np.abs(ys.cpu().numpy() - oracle.forward(xs.cpu().numpy(), w, np.float64, n_layers=1, n_layers_res=1)).max() < TOL
    assert len(ops._ENGINES) == 2
    with pytest.raises(ValueError):
        torch.ops.catfish.resnetrnn_forward(x.cpu(), packed)                                            # no CPU path
    with pytest.raises(ValueError):
        torch.ops.catfish.resnetrnn_forward(x.reshape(-1, 59), packed)                                  # windows are 35 samples
    ops.clear_engine_cache()
    assert not ops._ENGINES


def test_torch_operator_runs_the_weights_it_is_given_after_address_reuse(ckpt_weights):
    """pack A, run, delete, pack B of the same geometry into the memory A had (forced: ``empty_like`` + ``copy_``
    right after the free) -> B's probabilities, bit-identical to ``HipEngine(B)``; the old (address, version, numel) key returned A's."""
    import gc
    import torch
    import catfish_amd.torch_ops as ops
    from catfish_amd.engine import HipEngine
    ops.clear_engine_cache()
    wb = {k: np.array(v) for k, v in ckpt_weights.items()}
    wb["final_fully_connected/bias"] = wb["final_fully_connected/bias"] + np.float32(1.5)              # every probability moves
    x = torch.randn(64, 35, device="cuda")
    outs = {}
    for tag, w in (("a", ckpt_weights), ("b", wb)):
        e = HipEngine(w, device=0)
        outs[tag] = e.infer_device(x).clone()
        e.close()
    assert not torch.equal(outs["a"], outs["b"])
    template = ops.pack_weights(wb)
    same_address = 0
    for _ in range(5):
        a = ops.pack_weights(ckpt_weights)
        addr = a.untyped_storage().data_ptr()
        assert torch.equal(torch.ops.catfish.resnetrnn_forward(x, a), outs["a"])
        del a
        gc.collect()
        b = torch.empty_like(template)
        b.copy_(template)                                                                              # version 1, maybe A's address
        same_address += b.untyped_storage().data_ptr() == addr
        assert torch.equal(torch.ops.catfish.resnetrnn_forward(x, b), outs["b"])
        b2 = ops.pack_weights(wb)                                                                      # version 0, like A had
        assert torch.equal(torch.ops.catfish.resnetrnn_forward(x, b2), outs["b"])
        del b, b2
        gc.collect()
    assert len(ops._ENGINES) == 2
    ops.clear_engine_cache()
