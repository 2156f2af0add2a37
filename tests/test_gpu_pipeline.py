"""GPU tests of the read-level pipeline and of full-size properties (through the C ABI)."""
import json
import os

import numpy as np
import pytest

from oracle import catfish_oracle as oracle
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(hp):
    from catfish_amd.resnet_class import ResNetRNN
    m = ResNetRNN(**hp)
    with np.load(os.path.join(GOLDEN, "ckpnt-30000-inference.npz")) as z:
        m.set_weights({k: z[k] for k in z.files})
    yield m
    m.engine.close()


def test_model_infer_surface(model, golden_read):
    x = golden_read["x"][:, :, None].astype(np.float64)      # reference feeds float64 [N,35,1]
    out = model.infer(x)
    assert out.dtype == np.float64 and out.shape == (118 * 35,)
    assert np.abs(out - golden_read["probs_fp64"]).max() < 1e-4


def test_infer_class_from_raw_matches_oracle(model, ckpt_weights):
    from catfish_amd import infer
    for seed, n in ((1, 4096), (2, 700), (3, 35), (4, 36), (5, 1)):
        dac = oracle.synthetic_dac(1, max(n, 2), seed=seed)[0][:n]
        sig = oracle.normalize_raw_signal(dac) if n > 1 else np.array([0.3])
        spans, length = infer.infer_class_from_raw(sig, model)
        w_spans, w_len, _ = oracle.infer_read(sig, ckpt_weights, np.float32)
        assert length == w_len == n
        assert spans == w_spans


def test_infer_class_from_signal_npy_and_errors(model, tmp_path, ckpt_weights):
    from catfish_amd import infer
    dac = oracle.synthetic_dac(1, 2000, seed=9)[0]
    p = tmp_path / "read0.npy"
    np.save(p, dac)
    spans, length = infer.infer_class_from_signal(str(p), model)
    w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(dac), ckpt_weights, np.float32)
    assert (spans, length) == (w_spans, w_len)
    with pytest.raises(ValueError):
        infer.infer_class_from_signal(str(tmp_path / "missing.fast5"), model)


def test_test_network_counters(model, golden_read):
    x = golden_read["x"][:, :, None]
    y = (golden_read["probs_fp64"] >= 0.5).astype(np.float64).reshape(-1, 35, 1)
    acc, loss = model.test_network(x, y, "read", "", padding_size=34)
    assert acc == 1.0 and loss > 0
    assert model.tp + model.fn == int(y.sum())
    assert model.fp == 0 and model.fn == 0


def test_validation_round_is_one_packed_call_and_matches_the_oracle(model, ckpt_weights, tmp_path, monkeypatch):
    """train_validate.validate on the HIP engine: all reads of a round through ONE cf_infer_host_logits call; its
    report numbers against the same round scored read by read with the fp64 oracle's probabilities and logits."""
    from catfish_amd import train_validate as tv
    monkeypatch.chdir(tmp_path)
    paths = []
    for i, n in enumerate((1400, 735, 3000, 36, 2222, 700)):
        raw, lab = tv.synthetic_labelled_read(n, seed=70 + i)
        paths.append(str(tmp_path / ("sq%d.npz" % i)))
        np.savez(paths[-1], raw=raw, base_labels=lab)
    calls = []
    real = model.engine.infer_host
    monkeypatch.setattr(model.engine, "infer_host", lambda x, return_logits=False: (calls.append(np.shape(x)),
                                                                                     real(x, return_logits))[1])
    model.tp = model.fp = model.tn = model.fn = 0
    acc, precision, recall = tv.validate(model, paths, 0, "round", "complete", 5)
    assert len(calls) == 1 and calls[0] == (sum(-(-n // 35) for n in (1400, 735, 3000, 36, 2222)), 35, 1)
    assert (model.tp, model.fp, model.tn, model.fn) == (0, 0, 0, 0)
    # the same round from the oracle
    signals, labels = tv.select_validation_stretches(paths, 35, 0, "complete", 5)
    x, y, bounds, tails = tv.pack_validation_windows(signals, labels, 35)
    p64, st = oracle.forward(x, ckpt_weights, np.float64, return_stages=True)
    w_acc, w_loss, (tp, fp, tn, fn) = tv.score_validation_batch(p64.reshape(-1), st["logits"].reshape(-1), y, bounds, tails)
    assert tp + fp + tn + fn == 1400 + 735 + 3000 + 36 + 2222
    report = open("round.txt").read()
    assert "Detected {} true positives, {} false positives, {} true negatives, {} false negatives".format(tp, fp, tn, fn) in report
    assert "\tAccuracy: {:.2%}\n\tLoss: {:.4f}".format(float(np.sum(w_acc[:4].astype(float))) / 5,
                                                      float(np.sum(w_loss[:4].astype(float))) / 5) in report
    assert abs(acc - (tp + tn) / (tp + fp + tn + fn)) < 1e-12 and abs(precision - tp / max(tp + fp, 1)) < 1e-12
    assert abs(recall - tp / max(tp + fn, 1)) < 1e-12


def test_postprocess_kernel_matches_reference_semantics(model):
    torch = pytest.importorskip("torch")
    from catfish_amd import batching
    rng = np.random.default_rng(0)
    lens = [1, 14, 15, 16, 35, 36, 700, 4096, 70]
    sigs = [np.zeros(n) for n in lens]
    pk = batching.pack_reads(sigs)
    probs = np.zeros(pk.n_windows * 35, dtype=np.float32)
    per_read = []
    for i, n in enumerate(lens):
        base = np.cumsum(rng.normal(0, 0.15, size=n)) + rng.normal(0, 0.3)
        p = (1 / (1 + np.exp(-base))).astype(np.float32)
        per_read.append(p)
        probs[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = p
    # padding deliberately filled with ones: it must still come out as 0
    mask = np.ones_like(probs, dtype=bool)
    for i, n in enumerate(lens):
        mask[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = False
    probs[mask] = 1.0
    dev = torch.device("cuda", 0)
    labels = model.engine.postprocess_device(torch.from_numpy(probs).to(dev),
                                             torch.from_numpy(pk.sample_offsets).to(dev),
                                             torch.from_numpy(pk.lengths).to(dev)).cpu().numpy()
    assert labels[mask].sum() == 0
    for i, n in enumerate(lens):
        want = oracle.correct_short(oracle.class_from_threshold(per_read[i]))
        got = labels[pk.sample_offsets[i]:pk.sample_offsets[i] + n]
        assert np.array_equal(got, want), i
    spans = batching.spans_from_labels(labels, pk.sample_offsets, pk.n_reads)
    for i, n in enumerate(lens):
        want = oracle.correct_short(oracle.class_from_threshold(per_read[i]))
        assert spans[i] == (oracle.hp_in_pred(want) if want.any() else [])


def test_packed_variable_length_reads_match_per_read_oracle(model, ckpt_weights):
    """Config-4 shape: ragged reads 512..16384 in length-bucketed packed launches."""
    from catfish_amd import batching
    rng = np.random.default_rng(2)
    lens = np.exp(rng.uniform(np.log(512), np.log(16384), size=12)).astype(int).tolist() + [35, 70, 1]
    sigs = []
    for i, n in enumerate(lens):
        dac = oracle.synthetic_dac(1, max(n, 2), seed=100 + i)[0][:n]
        sigs.append(oracle.normalize_raw_signal(dac) if n > 1 else np.array([0.1]))
    got = batching.infer_reads(model, sigs, max_windows=1024)
    for s, g in zip(sigs, got):
        w_spans, w_len, _ = oracle.infer_read(s, ckpt_weights, np.float32)
        assert g == (w_spans, w_len)


def test_config4_bf16_packed_varlen(ckpt_weights):
    """BASELINE configs[3] as stated: variable-length reads 512..16384 (log-uniform, seed 2) as raw DAC squiggles, in
    length-bucketed PACKED launches, bf16 biGRU arithmetic -- through batching.infer_reads_dac (device normalisation,
    forward pass, device post-processing).  Judged like SURVEY 8d says, by the constants of oracle/tolerances.py that the
    bench's config4 leg uses too: label match rate against the fp32 oracle over all samples and per read, and a
    bound on max |dp| at this test's own scale (unpinned by the reference, see tolerances.py); edge lengths 512, 16384, 35k and 35k + 1 included."""
    from catfish_amd import batching
    from catfish_amd.engine import HipEngine
    from oracle import tolerances as tol
    rng = np.random.default_rng(2)
    lens = np.exp(rng.uniform(np.log(512), np.log(16384), size=22)).astype(int).tolist() + [512, 16384, 35 * 40, 35 * 40 + 1, 35 * 300]
    dacs = [oracle.synthetic_dac(1, n, seed=2000 + i)[0] for i, n in enumerate(lens)]
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=2048, precision="bf16")
    try:
        res, probs = batching.infer_reads_dac(eng, dacs, max_windows=2048, return_probs=True)   # several buckets
        # the fp32 engine on the same packed batches: spans of the two precisions must agree almost everywhere
    finally:
        eng.close()
    n_match = n_tot = 0
    worst = 0.0
    for d, (spans, n), p in zip(dacs, res, probs):
        assert n == len(d) and p.shape == (n,)
        x, pad = oracle.pad_and_window(oracle.normalize_raw_signal(d))
        want = oracle.forward(x, ckpt_weights, np.float32)[:n]
        worst = max(worst, float(np.abs(p - want).max()))
        m = (p >= 0.5) == (want >= 0.5)
        n_match += int(m.sum()); n_tot += n
        assert m.mean() >= tol.CONFIG4_MIN_LABEL_MATCH_PER_READ, (n, m.mean())
        # spans are what the tool emits: they come from the bf16 labels through correct_short + hp_in_pred
        lab = oracle.correct_short(oracle.class_from_threshold(p))
        assert spans == (oracle.hp_in_pred(lab) if lab.any() else [])
    print("config 4: label match %.5f over %d samples, max |dp| %.2e" % (n_match / n_tot, n_tot, worst))
    assert n_match / n_tot >= tol.CONFIG4_MIN_LABEL_MATCH
    assert worst <= tol.CONFIG4_MAX_ABS_DP_TEST          # the bound at THIS test's scale (1e-2), not the bench leg's looser one


def test_config4_at_scale_is_invariant_to_the_packing(ckpt_weights):
    """BASELINE configs[3] at a size the oracle cannot follow (2 000 reads, 9.3 M samples, bf16): windows are independent,
    so which reads share a launch must not matter.  The same reads through 4 096-window buckets, through the CLI's
    131 072-window launches (where the fused residual stack runs one chunk per tile and the biGRU kernels walk several
    tiles per wave) and read by read give bit-identical probabilities and identical spans; read lengths come back exact."""
    from catfish_amd import batching
    from catfish_amd.engine import HipEngine
    rng = np.random.default_rng(2)
    lens = np.rint(np.exp(rng.uniform(np.log(512), np.log(16384), size=2000))).astype(int).tolist()
    dacs = [np.clip(np.rint(rng.normal(500, 60, size=n)), 0, 2047).astype(np.int16) for n in lens]
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=131072, precision="bf16")
    try:
        small, p_small = batching.infer_reads_dac(eng, dacs, max_windows=4096, return_probs=True)
        big, p_big = batching.infer_reads_dac(eng, dacs, max_windows=131072, return_probs=True)
        assert [n for _s, n in small] == lens == [n for _s, n in big]
        assert small == big
        assert all(np.array_equal(a, b) for a, b in zip(p_small, p_big))
        for i in (0, 7, 1999, int(np.argmax(lens)), int(np.argmin(lens))):
            one, p_one = batching.infer_reads_dac(eng, [dacs[i]], max_windows=4096, return_probs=True)
            assert one[0] == big[i] and np.array_equal(p_one[0], p_big[i])
        assert all(np.isfinite(p).all() and p.min() >= 0 and p.max() <= 1 for p in p_big)
    finally:
        eng.close()


def test_full_size_properties(model):
    """BASELINE size (256 reads x 118 windows): window-permutation equivariance, batch-split
    invariance and run-to-run determinism -- size-independent properties of independent windows."""
    torch = pytest.importorskip("torch")
    eng = model.engine
    n = 256 * 118
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(n, 35, generator=g).mul_(1.5).cuda()
    a = eng.infer_device(x).clone()
    b = eng.infer_device(x).clone()
    assert torch.equal(a, b)                                   # deterministic
    perm = torch.randperm(n, generator=g).cuda()
    c = eng.infer_device(x[perm].contiguous()).view(n, 35)
    assert torch.equal(c, a.view(n, 35)[perm])                 # windows independent, bit-exact
    half = eng.infer_device(x[: n // 2 + 7].contiguous())
    assert torch.equal(half, a[: (n // 2 + 7) * 35])           # split point does not matter
    assert torch.isfinite(a).all() and (a > 0).all() and (a < 1).all()
    # spot-check 32 random windows of the big batch against the oracle
    idx = torch.randint(0, n, (32,), generator=g)
    with np.load(os.path.join(GOLDEN, "ckpnt-30000-inference.npz")) as z:
        w = {k: z[k] for k in z.files}
    want = oracle.forward(x[idx.cuda()].cpu().numpy(), w, np.float64).reshape(32, 35)
    got = a.view(n, 35)[idx.cuda()].cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


def test_bf16_full_size_properties(ckpt_weights):
    """BASELINE configs[3]'s dtype at the benchmark's launch size (30 208 windows, the software-pipelined bf16 kernel):
    run-to-run determinism, window-permutation equivariance and batch-split invariance, bit-exact (a window's arithmetic
    does not depend on its tile-mates), plus a sample against the fp64 oracle at the bf16 tolerance."""
    torch = pytest.importorskip("torch")
    from catfish_amd.engine import HipEngine
    n = 256 * 118
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=n, precision="bf16")
    try:
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(n, 35, generator=g).mul_(1.5).cuda()
        a = eng.infer_device(x).clone()
        assert torch.equal(a, eng.infer_device(x))
        perm = torch.randperm(n, generator=g).cuda()
        c = eng.infer_device(x[perm].contiguous()).view(n, 35)
        assert torch.equal(c, a.view(n, 35)[perm])
        part = eng.infer_device(x[: n // 2 + 13].contiguous())
        assert torch.equal(part, a[: (n // 2 + 13) * 35])
        assert torch.isfinite(a).all() and (a >= 0).all() and (a <= 1).all()
        idx = torch.randint(0, n, (48,), generator=g)
        want = oracle.forward(x[idx.cuda()].cpu().numpy(), ckpt_weights, np.float64).reshape(48, 35)
        got = a.view(n, 35)[idx.cuda()].cpu().numpy()
        assert np.abs(got - want).max() < 3e-2 and np.mean((got >= 0.5) == (want >= 0.5)) > 0.99
        eng.check_error()
    finally:
        eng.close()


def test_logits_output_and_pipeline_options(model, ckpt_weights):
    """cf_infer_logits on device buffers: sigmoid(logits) == probs; the streaming pipeline gives the same spans with three
    batches in flight and with every kernel of a batch on the compute stream (the two knobs of ReadPipeline)."""
    torch = pytest.importorskip("torch")
    from catfish_amd.pipeline import ReadPipeline
    x = torch.randn(300, 35, device="cuda")
    logits = torch.empty(300 * 35, device="cuda")
    probs = model.engine.infer_device(x, logits=logits)
    assert torch.allclose(torch.sigmoid(logits), probs, atol=1e-6)
    assert torch.equal(probs, model.engine.infer_device(x))
    lens = [4096, 700, 35, 5000, 36, 2048, 999, 1234, 4096, 512]
    dacs = [oracle.synthetic_dac(1, n, seed=800 + i)[0] for i, n in enumerate(lens)]
    batches = [dacs[0:3], dacs[3:5], dacs[5:8], dacs[8:10], dacs[2:6]]
    ref = [r for res in ReadPipeline(model.engine, 12000).run(batches) for r in res]
    for kw in (dict(depth=3), dict(overlap_kernels=False), dict(depth=4, overlap_kernels=False)):
        got = [r for res in ReadPipeline(model.engine, 12000, **kw).run(batches) for r in res]
        assert got == ref, kw
    w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(dacs[0]), ckpt_weights, np.float32)
    assert ref[0] == (w_spans, w_len)


def test_extreme_inputs_saturate_cleanly(model):
    x = np.zeros((32, 35), np.float32)
    x[0] = 1e4; x[1] = -1e4; x[2, ::2] = 300; x[3] = np.linspace(-50, 50, 35)
    out = model.engine.infer_host(x)
    assert np.isfinite(out).all()
    with np.load(os.path.join(GOLDEN, "ckpnt-30000-inference.npz")) as z:
        w = {k: z[k] for k in z.files}
    want = oracle.forward(x, w, np.float64)
    assert np.abs(out - want).max() < 1e-4


def test_device_normalisation_is_bit_exact(model):
    """cf_normalize vs numpy float64 normalisation cast to float32 (infer.py:96-105), incl. padding."""
    torch = pytest.importorskip("torch")
    from catfish_amd import batching, infer
    rng = np.random.default_rng(4)
    lens = [1, 2, 3, 34, 35, 36, 70, 999, 4096, 10001]
    reads = [np.clip(np.rint(rng.normal(500, 60, size=n)), 0, 2047).astype(np.int16) for n in lens]
    reads.append(rng.integers(-32768, 32767, size=777).astype(np.int16))      # full int16 range
    reads.append(np.array([5, 5, 5, 9, 1, 5, 5, 5, 5, 7], dtype=np.int16))     # many ties
    lens = [len(r) for r in reads]
    dev = torch.device("cuda", 0)
    dac_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    n_win = [(n + infer.padding_size_for(n)) // 35 for n in lens]
    win_off = np.concatenate(([0], np.cumsum(n_win))).astype(np.int64)
    x = model.engine.normalize_device(torch.from_numpy(np.concatenate(reads)).to(dev),
                                      torch.from_numpy(dac_off).to(dev), torch.from_numpy(win_off).to(dev)).cpu().numpy()
    for i, r in enumerate(reads):
        with np.errstate(divide="ignore", invalid="ignore"):
            want = infer.normalize_raw_signal(r, "median").astype(np.float32)
        got = x[win_off[i]:win_off[i + 1]].reshape(-1)
        assert np.array_equal(got[:lens[i]], want, equal_nan=True), i
        assert np.all(got[lens[i]:] == 0), i


def test_dac_pipeline_matches_oracle(model, ckpt_weights):
    from catfish_amd import batching
    dacs = [oracle.synthetic_dac(1, n, seed=40 + n)[0] for n in (512, 4096, 700, 9000)]
    got = batching.infer_reads_dac(model, dacs, max_windows=2048)
    for d, g in zip(dacs, got):
        w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        assert g == (w_spans, w_len)


def test_train_network_then_infer_uses_updated_weights(hp):
    """Config 5 surface: train_network (torch-ROCm autograd, Adam) then infer through the HIP engine."""
    pytest.importorskip("torch")
    from catfish_amd.resnet_class import ResNetRNN
    m = ResNetRNN(**dict(hp, optimizer_choice="Adam", train_seed=0))
    m.set_weights(oracle.random_weights(seed=9))
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1.2, size=(256, 35, 1)).astype(np.float32)
    y = np.repeat((np.arange(256) % 2)[:, None], 35, axis=1).astype(np.float32)[:, :, None]
    before = m.infer(x[:32])
    losses = []
    for step in range(3):
        m.train_network(x, y, step)
        losses.append(m.train_loss)
    after = m.infer(x[:32])                       # engine re-tiled from the trained weights
    assert np.isfinite(losses).all() and not np.allclose(before, after)
    want = oracle.forward(x[:32], m._trainer.net.numpy_weights(), np.float64)
    assert np.abs(after - want).max() < 1e-4
    m.engine.close()


def test_train_save_load_infer_loop(hp, tmp_path, monkeypatch):
    """The loop the reference's training script closes (networks/train_validate.py:323-334,154 ->
    neural_network.load_network, neural_network.py:26-34): build with save=True (model directory + report),
    train 20 steps on balanced synthetic batches, checkpoint, validate on NPZ reads, then load the directory back
    with load_network and get the same predictions as the trained model (and as the oracle on the saved weights)."""
    pytest.importorskip("torch")
    from catfish_amd import neural_network, train_validate as tv, checkpoint
    monkeypatch.chdir(tmp_path)
    net = tv.build_model("ResNetRNN", save=True, **dict(hp, batch_size=64, train_seed=0))
    net.initialize_network(seed=4)
    db = tv.synthetic_example_db(n_reads=2, read_len=12000, seed=2)
    val_dir = tmp_path / "val"
    val_dir.mkdir()
    squiggles = []
    for i in range(3):
        raw, lab = tv.synthetic_labelled_read(3000 + 35 * i, seed=50 + i)
        np.savez(val_dir / ("sq%d.npz" % i), raw=raw, base_labels=lab)
        squiggles.append(str(val_dir / ("sq%d.npz" % i)))
    acc = tv.train_and_validate(net, db, 20 * 64, squiggles, 2000, net.model_path, 0, 856)
    assert 0.0 <= acc <= 1.0 and np.isfinite(net.train_loss)
    report = open(net.model_path + ".txt").read()
    assert "Training on 1280 examples in 20 batches" in report and "Saved checkpoint at step 20" in report
    val_report = open(os.path.basename(net.model_path) + ".txt").read()       # validate() writes <basename>.txt in the CWD
    assert "---NEXT ROUND OF VALIDATION---" in val_report and "F1 score" in val_report
    assert (net.tp, net.fp, net.tn, net.fn) == (0, 0, 0, 0)                    # counters reset after validation
    # load the model directory back like the inference tool does
    loaded = neural_network.load_network("ResNetRNN", net.model_path, checkpoint=20)
    x = np.random.default_rng(0).normal(0, 1.2, size=(40, 35, 1))
    got = loaded.infer(x)
    assert np.array_equal(got, net.infer(x))
    saved = checkpoint.read_inference_weights(os.path.join(net.model_path, "checkpoints"), "ckpnt-20")
    assert np.abs(got - oracle.forward(x, saved, np.float64)).max() < 1e-4
    assert loaded.optimizer_state and any(k.endswith("/RMSProp") for k in loaded.optimizer_state)
    # evaluate(): accuracy / logits-based loss of a batch against the oracle's logits
    y = (np.arange(40 * 35).reshape(40, 35, 1) % 3 == 0).astype(np.float64)
    acc2, loss2 = loaded.evaluate(x, y)
    _, st = oracle.forward(x, saved, np.float64, return_stages=True)
    z = st["logits"].reshape(-1)
    want_loss = np.mean(np.maximum(z, 0) - z * y.reshape(-1) + np.log1p(np.exp(-np.abs(z))))
    assert abs(loss2 - want_loss) < 1e-4 and abs(acc2 - np.mean((z >= 0) == (y.reshape(-1) == 1))) < 1e-3
    loaded.engine.close(); net.engine.close()


@pytest.mark.parametrize("network_type,hpm", [
    ("ResNetRNN", dict(batch_size=64, optimizer_choice="Adam", learning_rate=0.001, layer_size=128, n_layers=2, keep_prob=0.7,
                       layer_size_res=64, n_layers_res=2)),
    ("RNN", dict(batch_size=64, optimizer_choice="RMSProp", learning_rate=0.001, layer_size=32, n_layers=1, keep_prob=0.8))])
def test_other_hyperparameter_draws_train_save_load_infer(network_type, hpm, tmp_path, monkeypatch):
    """A draw of the reference's hyper-parameter search other than the shipped geometry (networks/train_validate.py:66-111,
    :328): build with save=True, train a few steps (torch autograd around the any-size HIP recurrence kernels; the fully native
    step is built for 64 / 32), checkpoint, load the model directory back with load_network and infer on the any-size HIP kernels:
    same predictions as the trained object and as the oracle on the saved weights, and the loss went down."""
    pytest.importorskip("torch")
    from catfish_amd import neural_network, train_validate as tv, checkpoint
    monkeypatch.chdir(tmp_path)
    net = tv.build_model(network_type, save=True, **dict(hpm, train_seed=0))
    net.initialize_network(seed=6)
    assert net.engine.launch_regimes()["coop_max"] == 0                      # the any-size path
    db = tv.synthetic_example_db(n_reads=2, read_len=12000, seed=3)
    losses = []
    for step in range(12):
        data, labels, _ = db.get_training_set(64, ratio=2)
        net.train_network(tv.reshape_input(data, 35, 1), tv.reshape_input(labels, 35, 1), step + 1)
        losses.append(net.train_loss)
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < np.mean(losses[:3])
    net.save_network_to_model_path(12)
    loaded = neural_network.load_network(network_type, net.model_path, checkpoint=12)
    assert (loaded.layer_size, loaded.n_layers) == (hpm["layer_size"], hpm["n_layers"])
    x = np.random.default_rng(0).normal(0, 1.2, size=(50, 35, 1))
    got = loaded.infer(x)
    assert np.array_equal(got, net.infer(x))
    saved = checkpoint.read_inference_weights(os.path.join(net.model_path, "checkpoints"), "ckpnt-12")
    want = oracle.forward(x, saved, np.float64, n_layers=hpm["n_layers"], n_layers_res=hpm.get("n_layers_res", 0))
    assert np.abs(got - want).max() < 1e-4
    loaded.engine.close(); net.engine.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("network_type,seed", [("ResNetRNN", 0), ("ResNetRNN", 3), ("RNN", 1), ("ResNetRNN", 11)])
def test_train_validate_script_with_the_reference_default_random_draw(network_type, seed, tmp_path, monkeypatch):
    """`python -m catfish_amd.train_validate <type> <train dir> <n> <val dir> <len> <start>` as the reference runs it
    (networks/train_validate.py:299-347): hyper-parameters drawn at random (:328) -- whatever geometry comes up builds,
    trains two batches, checkpoints, validates and writes both reports; the model directory loads back."""
    pytest.importorskip("torch")
    from catfish_amd import neural_network, train_validate as tv
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("CATFISH_SHIPPED_HPARAMS", raising=False)
    (tmp_path / "train").mkdir()
    (tmp_path / "val").mkdir()
    for i in range(2):
        raw, lab = tv.synthetic_labelled_read(30000, seed=70 + i)
        np.savez(tmp_path / "train" / ("t%d.npz" % i), raw=raw, base_labels=lab)
        raw, lab = tv.synthetic_labelled_read(3000, seed=80 + i)
        np.savez(tmp_path / "val" / ("v%d.npz" % i), raw=raw, base_labels=lab)
    np.random.seed(seed)
    hpm = tv.generate_random_hyperparameters(network_type)                  # what main() is about to draw
    np.random.seed(seed)
    tv.main(["train_validate.py", network_type, str(tmp_path / "train"), str(2 * hpm["batch_size"]), str(tmp_path / "val"), "1050", "0"])
    (model_dir,) = [d for d in tmp_path.iterdir() if d.is_dir() and d.name.endswith("_0")]      # <model type>_0, rnn_class.py:100-118
    report = open(str(model_dir) + ".txt").read()
    assert "layer_size: %d" % hpm["layer_size"] in report and "Saved checkpoint at step 2" in report
    loaded = neural_network.load_network(network_type, str(model_dir), checkpoint=2)
    assert loaded.layer_size == hpm["layer_size"] and loaded.n_layers == hpm["n_layers"]
    out = loaded.infer(np.zeros((3, 35, 1)))
    assert out.shape == (105,) and np.isfinite(out).all()
    loaded.engine.close()


def test_plain_rnn_type_matches_oracle(hp):
    """build_model("RNN") (neural_network.py:17-18): 3 x biGRU directly on the raw window, no residual blocks."""
    from catfish_amd import neural_network
    m = neural_network.build_model("RNN", **hp)
    assert m.model_type == "biGRU-RNN"
    w = oracle.random_weights(seed=21, n_layers_res=0)
    m.set_weights(w)
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1.3, size=(77, 35, 1))
    got = m.infer(x)
    want = oracle.forward(x, w, np.float64, n_layers_res=0)
    assert np.abs(got - want).max() < 1e-4
    stage = m.engine.debug_stage(0, 16)
    _, st = oracle.forward(x[:16], w, np.float64, n_layers_res=0, return_stages=True)
    assert np.abs(stage - st["gru0"]).max() < 2e-5
    m.engine.close()
    with pytest.raises(ValueError):
        from catfish_amd.engine import HipEngine
        HipEngine(w, n_layers_res=0, precision="bf16")


def test_save_network_round_trip(model, tmp_path, hp):
    from catfish_amd.resnet_class import ResNetRNN
    prefix = model.save_network(str(tmp_path), 123)
    assert prefix.endswith("ckpnt-123")
    m2 = ResNetRNN(**hp)
    m2.restore_network(str(tmp_path), ckpnt="latest")
    x = np.random.default_rng(0).normal(size=(20, 35, 1))
    assert np.array_equal(m2.infer(x), model.infer(x))
    m2.engine.close()


def test_streaming_pipeline_matches_oracle(model, ckpt_weights):
    """ReadPipeline: pinned int16 DAC in -> spans out, double-buffered; results equal the per-read oracle."""
    from catfish_amd.pipeline import ReadPipeline
    lens = [4096, 700, 35, 5000, 36, 2048, 999, 1234]
    dacs = [oracle.synthetic_dac(1, n, seed=300 + i)[0] for i, n in enumerate(lens)]
    pipe = ReadPipeline(model.engine, max_samples_per_batch=12000)
    batches = [dacs[0:3], dacs[3:5], dacs[5:8]]
    got = [r for res in pipe.run(batches) for r in res]
    for d, g in zip(dacs, got):
        w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        assert g == (w_spans, w_len)
    # array form carries the same information
    t = pipe.submit(batches[0])
    read_of, s, e, ln = pipe.collect(t, as_lists=False)
    flat = [[int(a), int(b)] for a, b in zip(s, e)]
    assert flat == [sp for spans, _ in got[:3] for sp in spans] and list(ln) == lens[:3]
    with pytest.raises(ValueError):
        pipe.submit([np.zeros(7000, np.int16), np.zeros(7000, np.int16)])     # several reads beyond the batch size
    # a batch whose results are bad (here: run counts that disagree) frees its slot; the pipeline keeps working
    for _ in range(pipe.depth + 1):
        t = pipe.submit(batches[0])
        t.done.synchronize()
        t.counts_h[0] += 1
        with pytest.raises(RuntimeError, match="starts"):
            pipe.collect(t)
        assert t.keep is None and all(x is None for x in pipe.inflight)
    assert [r for res in pipe.run(batches) for r in res] == got
    model.engine.clear_error()                                 # no error pending: a no-op that must succeed
    model.engine.check_error()
    # ONE read longer than the batch size grows the staging buffers instead (a directory may hold such a read)
    long_dac = oracle.synthetic_dac(1, 20000, seed=77)[0]
    (spans, n), = pipe.collect(pipe.submit([long_dac]))
    w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(long_dac), ckpt_weights, np.float32)
    assert (spans, n) == (w_spans, w_len)


def test_any_size_model_through_the_streaming_pipeline():
    """The host-to-host pipeline (device normalisation, forward pass, post-processing, spans) with a 32-unit / 16-channel
    model on the any-size kernels: spans equal the oracle's per read."""
    from catfish_amd.engine import HipEngine
    from catfish_amd.pipeline import ReadPipeline
    w = oracle.random_weights(seed=77, layer_size=32, layer_size_res=16)
    w["final_fully_connected/bias"] = np.array([0.35], np.float32)           # random weights: push some samples over 0.5
    eng = HipEngine(w, layer_size=32, n_layers=3, layer_size_res=16, n_layers_res=2, device=0, max_windows_per_pass=4096)
    try:
        lens = [4096, 700, 35, 5000, 36, 2048]
        dacs = [oracle.synthetic_dac(1, n, seed=900 + i)[0] for i, n in enumerate(lens)]
        got = [r for res in ReadPipeline(eng, 12000).run([dacs[:3], dacs[3:]]) for r in res]
        for d, g in zip(dacs, got):
            spans, n, _ = oracle.infer_read(oracle.normalize_raw_signal(d), w, np.float32)
            assert g == (spans, n)
        eng.check_error()
    finally:
        eng.close()


def test_ultra_long_read_through_the_pipeline(model, tmp_path):
    """One 3 000 017-sample read (ultra-long nanopore reads are millions of samples): 85 715 windows, more than the
    engine's launch capacity, so cf_infer runs it in several passes; the staging buffers grow; one workgroup finds the
    exact medians of 3 M samples.  The device route (int16 -> cf_normalize -> cf_infer -> cf_postprocess -> cf_spans)
    must give the same spans as the host route of infer_class_from_signal (numpy normalisation in float64, cf_infer_host,
    host post-processing) on the same read."""
    from catfish_amd import infer as cinfer
    from catfish_amd.pipeline import ReadPipeline
    n = 3000017
    rng = np.random.default_rng(5)
    level = np.repeat(rng.normal(500, 60, size=n // 9 + 1), 9)[:n]
    dac = np.clip(level + rng.normal(0, 8, size=n), 0, 2047).astype(np.int16)
    path = tmp_path / "long.npy"
    np.save(path, dac)
    want_spans, want_len = cinfer.infer_class_from_signal(str(path), model)
    assert want_len == n
    pipe = ReadPipeline(model.engine, max_samples_per_batch=100000)
    (spans, length), = pipe.collect(pipe.submit([dac]))
    assert length == n and spans == want_spans and len(spans) > 100
    model.engine.check_error()


@pytest.mark.parametrize("h,c,n_layers,n_layers_res,n", [(32, 16, 2, 1, 40), (128, 64, 1, 1, 150), (48, 0, 2, 0, 75),
                                                         (256, 32, 1, 1, 33), (16, 80, 3, 2, 600), (64, 64, 2, 1, 100)])
def test_any_size_training_kernels_match_torch_autograd(h, c, n_layers, n_layers_res, n):
    """cf_gru_anysize_train_forward / _backward (+ library GEMMs for dx and dW, catfish_amd/anysize_train.py) against torch
    autograd of the restated graph at geometries other than 64 / 32: same loss, same gradient for every parameter, with
    explicit output-dropout masks; ragged window counts, the plain RNN type and a multi-workgroup batch included."""
    torch = pytest.importorskip("torch")
    from catfish_amd.training import TorchResNetRNN
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=40 + h, layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res)
    rng = np.random.default_rng(h + c)
    x = rng.normal(0, 1.2, size=(n, 35)).astype(np.float32)
    y = np.repeat((np.arange(n) % 2)[:, None], 35, axis=1).astype(np.float32)
    masks = {(l, d): (rng.random((n, 35, h)) < 0.8).astype(np.float32) for l in range(n_layers) for d in ("fw", "bw")}
    ref = TorchResNetRNN(w, n_layers, n_layers_res, device="cuda")
    nat = TorchResNetRNN(w, n_layers, n_layers_res, device="cuda")
    eng = HipEngine(w, layer_size=h, n_layers=n_layers, layer_size_res=max(c, 16), n_layers_res=n_layers_res, device=0,
                    max_windows_per_pass=256)
    try:
        l_ref = ref.loss(x, y, keep_prob=0.8, masks=masks)
        l_ref.backward()
        l_nat = nat.loss(x, y, keep_prob=0.8, engine=eng, masks=masks)
        l_nat.backward()
        assert abs(float(l_ref.detach()) - float(l_nat.detach())) < 1e-5
        for k, p_ref in ref.trainable().items():
            g_ref, g_nat = p_ref.grad, nat.trainable()[k].grad
            assert g_nat is not None, k
            scale = float(g_ref.abs().max()) + 1e-6
            assert float((g_ref - g_nat).abs().max()) < 2e-4 * scale + 1e-6, (k, scale)
    finally:
        eng.close()


def test_trainer_uses_the_any_size_kernels_for_other_geometries():
    """Trainer on a 128 / 64 model: the recurrence runs on the any-size HIP kernels (trainer.anysize); the gradients of a
    batch equal the pure-torch trainer's, and ten Adam steps under HIP-graph replay follow its loss trajectory (Adam
    normalises the gradient, so rounding-level differences in near-zero gradients move individual weights by up to the
    learning rate: the trajectory is compared, not the weights)."""
    torch = pytest.importorskip("torch")
    from catfish_amd.training import Trainer
    w = oracle.random_weights(seed=5, layer_size=128, n_layers=2, layer_size_res=64, n_layers_res=1)
    rng = np.random.default_rng(0)
    a = Trainer(w, 2, 1, "Adam", 1e-3, 1.0, seed=1)
    b = Trainer(w, 2, 1, "Adam", 1e-3, 1.0, seed=1, native=False)
    assert a.anysize and a.engine is not None and not a.native and not b.anysize
    x = rng.normal(0, 1.0, size=(64, 35)).astype(np.float32)
    y = np.repeat((rng.random(64) < 0.5)[:, None], 35, axis=1).astype(np.float32)
    la0, ga = a.gradients(x, y)
    lb0, gb = b.gradients(x, y)
    assert abs(la0 - lb0) < 1e-6
    for k in gb:
        assert np.abs(ga[k] - gb[k]).max() < 2e-4 * np.abs(gb[k]).max() + 1e-7, k
    la, lb = [], []
    for _ in range(10):
        x = rng.normal(0, 1.0, size=(64, 35)).astype(np.float32)
        y = np.repeat((rng.random(64) < 0.5)[:, None], 35, axis=1).astype(np.float32)
        la.append(a.train_step(x, y))
        lb.append(b.train_step(x, y))
    assert np.isfinite(la).all() and np.allclose(la, lb, rtol=0, atol=2e-3), (la, lb)
    assert la[-1] < la[0]
    a.engine.close()


@pytest.mark.parametrize("n", [40, 2100])
def test_native_gru_training_kernels_match_torch_autograd(n):
    """cf_gru_train_forward/backward (+ library GEMMs for dW) against torch autograd of the restated graph:
    same loss, same gradients for every parameter and for the input.  40 windows run the four-waves-per-tile
    (latency) kernels, 2100 windows (2 x 132 tiles > 256 CUs) the one-wave-per-tile (throughput) kernels."""
    torch = pytest.importorskip("torch")
    from catfish_amd.training import TorchResNetRNN
    from catfish_amd.engine import HipEngine
    w = oracle.random_weights(seed=31)
    rng = np.random.default_rng(2)
    x = rng.normal(0, 1.2, size=(n, 35)).astype(np.float32)           # not a multiple of 16: exercises the padding
    y = np.repeat((np.arange(n) % 2)[:, None], 35, axis=1).astype(np.float32)
    ref = TorchResNetRNN(w, 3, 2, device="cuda")
    nat = TorchResNetRNN(w, 3, 2, device="cuda")
    eng = HipEngine(w, device=0, max_windows_per_pass=max(256, n), fuse_layers=False)
    try:
        l_ref = ref.loss(x, y)
        l_ref.backward()
        l_nat = nat.loss(x, y, engine=eng)
        l_nat.backward()
        assert abs(float(l_ref.detach()) - float(l_nat.detach())) < 1e-5
        for k, p in ref.params.items():
            if p.grad is None:
                continue
            g = nat.params[k].grad
            assert g is not None, k
            scale = float(p.grad.abs().max()) + 1e-8
            err = float((p.grad - g).abs().max()) / scale
            assert err < 2e-3, (k, err)
    finally:
        eng.close()


@pytest.mark.parametrize("n,n_blocks", [(41, 2), (3, 1), (130, 3)])
def test_native_res_stack_training_kernels_match_torch_autograd(n, n_blocks):
    """cf_res_train_forward/backward (resnet_class.py:44-82 and its gradients) against torch autograd of the
    restated blocks: output and every kernel / bias / gamma / beta gradient; ragged window counts."""
    torch = pytest.importorskip("torch")
    from catfish_amd.training import TorchResNetRNN
    from catfish_amd.engine import HipEngine
    from catfish_amd.native_train import native_res_stack, res_unit_names
    w = oracle.random_weights(seed=17, n_layers=1, n_layers_res=n_blocks)
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.normal(0, 1.2, size=(n, 35)).astype(np.float32)).cuda()
    net = TorchResNetRNN(w, 1, n_blocks, device="cuda")
    eng = HipEngine(w, n_layers=1, n_layers_res=n_blocks, device=0, max_windows_per_pass=256, fuse_layers=False)
    try:
        a = x[:, None, :]
        for d in range(n_blocks):
            sc = net._conv_bn(a, 4 * d)
            o = torch.relu(net._conv_bn(a, 4 * d + 1))
            o = torch.relu(net._conv_bn(o, 4 * d + 2))
            o = torch.relu(net._conv_bn(o, 4 * d + 3))
            a = torch.relu(o + sc)
        ref = a.permute(0, 2, 1)
        names = [k for unit in res_unit_names(n_blocks) for k in unit]
        got = native_res_stack(x, [net.params[k] for k in names], eng)
        assert float((got - ref).detach().abs().max()) < 1e-5 * max(1.0, float(ref.detach().abs().max()))
        g = torch.randn_like(ref)
        train = [k for k in names if net.params[k].requires_grad]
        gr_ref = torch.autograd.grad((ref * g).sum(), [net.params[k] for k in train], retain_graph=True)
        gr_got = torch.autograd.grad((got * g).sum(), [net.params[k] for k in train])
        for k, r, h in zip(train, gr_ref, gr_got):
            assert float((r - h).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-6, k
    finally:
        eng.close()


def test_native_trainer_follows_the_torch_trainer():
    pytest.importorskip("torch")
    from catfish_amd.training import Trainer
    w = oracle.random_weights(seed=8)
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1.2, size=(256, 35)).astype(np.float32)
    y = np.repeat((np.arange(256) % 2)[:, None], 35, axis=1).astype(np.float32)
    a = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=1.0, device="cuda", native=True)
    b = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=1.0, device="cuda", native=False, use_graph=False)
    la = [a.train_step(x, y) for _ in range(5)]
    lb = [b.train_step(x, y) for _ in range(5)]
    assert np.allclose(la, lb, rtol=0, atol=2e-4), (la, lb)
    assert la[-1] < la[0]
    a.engine.close()


def test_native_trainer_follows_the_reference_training_graph(ckpt_weights):
    """Three RMSProp steps on the HIP training kernels (fp32, graph-captured) against the same three steps of the
    reference's own TF graph (tests/golden/graph_train_golden.npz from make_graph_golden.py; exact arithmetic per
    step, variables stored as float32 between steps like TF's): per-step loss and the accumulated change of all 58
    trainable variables."""
    import os
    pytest.importorskip("torch")
    from catfish_amd.training import Trainer
    from conftest import GOLDEN
    with np.load(os.path.join(GOLDEN, "graph_train_golden.npz")) as z:
        g = {k: z[k] for k in z.files}
    tr = Trainer(ckpt_weights, 3, 2, "RMSProp", 1e-3, keep_prob=1.0, device="cuda", native=True)
    start = {k: v.detach().clone() for k, v in tr.net.trainable().items()}
    for step in range(g["train_x"].shape[0]):
        loss = tr.train_step(g["train_x"][step], g["train_y"][step])
        assert abs(loss - float(g["train32_loss"][step])) < 5e-6, (step, loss)
    for k, p in tr.net.trainable().items():
        delta = (p.detach() - start[k]).cpu().numpy()
        ref = g["train32_delta/" + k]
        # an update that lands near a rounding boundary may round the other way: one ulp(|p|) per step
        slack = 3e-3 * np.abs(ref).max() + 3 * np.finfo(np.float32).eps * float(start[k].abs().max()) + 1e-8
        assert np.abs(delta - ref).max() <= slack, (k, np.abs(delta - ref).max(), np.abs(ref).max())
    tr.engine.close()


def test_native_training_with_dropout_matches_the_reference_graph(ckpt_weights):
    """keep_prob 0.8 with the masks the reference graph drew (graph_train_golden.npz, drop_*): loss and gradients
    of the HIP training kernels + torch remainder against the interpreted reference graph."""
    import os
    torch = pytest.importorskip("torch")
    from catfish_amd.training import TorchResNetRNN
    from catfish_amd.engine import HipEngine
    from conftest import GOLDEN
    with np.load(os.path.join(GOLDEN, "graph_train_golden.npz")) as z:
        g = {k: z[k] for k in z.files if k.startswith("drop_")}
    masks = {(layer, d): g["drop_masks"][layer, di] for layer in range(3) for di, d in enumerate(("fw", "bw"))}
    net = TorchResNetRNN(ckpt_weights, 3, 2, device="cuda")
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=256, fuse_layers=False)
    try:
        loss = net.loss(g["drop_x"], g["drop_y"], keep_prob=float(g["drop_keep_prob"]), engine=eng, masks=masks)
        loss.backward()
        assert abs(float(loss.detach()) - float(g["drop_loss"])) < 5e-6
        for k in [n[len("drop_grad/"):] for n in g if n.startswith("drop_grad/")]:
            ref = g["drop_grad/" + k]
            got = net.params[k].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), k
    finally:
        eng.close()


def test_train_head_kernel_matches_torch():
    """cf_train_head (dense 128 -> 1 + sigmoid cross-entropy on the logits, forward and backward in one pass over the
    fragment layout; rnn_class.py:178-183,74-79) against torch: loss, d kernel, d bias, d input, logits; a ragged batch
    (padding windows must contribute nothing)."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from catfish_amd import _native as N
    from catfish_amd.engine import HipEngine
    from catfish_amd.native_train import nat_to_frag, frag_to_nat
    eng = HipEngine(oracle.random_weights(seed=1), device=0, max_windows_per_pass=256)
    try:
        for n in (256, 37, 3000):
            g = torch.Generator(device="cpu").manual_seed(n)
            npad = (n + 15) // 16 * 16
            hin = (torch.rand(npad, 35, 128, generator=g) * 2 - 1).cuda()
            w = (torch.randn(128, generator=g) * 0.3).cuda()
            b = torch.tensor([0.17]).cuda()
            y = (torch.rand(n, 35, generator=g) < 0.3).float().cuda()
            yf = nat_to_frag(hin)
            dy = torch.empty_like(yf)
            logits = torch.zeros(n, 35, device="cuda")
            ws = torch.empty(int(eng._lib.cf_train_head_workspace_floats(eng._handle, npad)), device="cuda")
            grads = torch.zeros(129, device="cuda")
            loss = torch.zeros(1, device="cuda")
            p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
            N.check(eng._lib.cf_train_head(eng._handle, p(yf), p(w), p(b), p(y), n, p(dy), p(logits), p(ws), ws.numel(), p(grads), p(loss),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            href = hin[:n].clone().requires_grad_(True)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            z = href @ wr + br
            lref = torch.nn.functional.binary_cross_entropy_with_logits(z, y, reduction="mean")
            lref.backward()
            assert abs(float(loss) - float(lref)) < 2e-6
            assert torch.allclose(logits, z.detach(), atol=2e-5)
            assert torch.allclose(grads[:128], wr.grad, atol=1e-6, rtol=1e-4) and abs(float(grads[128]) - float(br.grad)) < 1e-6
            dnat = frag_to_nat(dy)
            assert torch.allclose(dnat[:n], href.grad, atol=1e-9, rtol=1e-4)
            assert float(dnat[n:].abs().max()) == 0.0 if npad > n else True
    finally:
        eng.close()


def test_in_kernel_dropout_matches_torch_with_the_same_masks(ckpt_weights):
    """The native training step draws its output-dropout masks INSIDE the biGRU kernels (a hash of seed, layer, optimizer
    step and element index; no mask tensor).  cf_dropout_scale writes the same factors out; replayed through the
    torch-autograd restatement of the graph they must give the same loss and gradients.  The masks keep ~keep_prob of
    the elements, differ per layer and change with the optimizer step."""
    torch = pytest.importorskip("torch")
    from catfish_amd.training import Trainer, TorchResNetRNN
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1.2, size=(64, 35)).astype(np.float32)
    y = np.repeat((np.arange(64) % 2)[:, None], 35, axis=1).astype(np.float32)
    tr = Trainer(ckpt_weights, 3, 2, "RMSProp", 1e-3, keep_prob=0.8, device="cuda", native=True, seed=5, use_graph=False)
    try:
        masks = tr.step_impl.dropout_scales(64)
        frac = np.mean([m.mean() for m in masks.values()])
        assert abs(frac - 0.8) < 0.01
        assert not np.array_equal(masks[(0, "fw")], masks[(1, "fw")]) and not np.array_equal(masks[(0, "fw")], masks[(0, "bw")])
        loss_n, grads_n = tr.gradients(x, y)                               # in-kernel masks, no update
        ref = TorchResNetRNN(ckpt_weights, 3, 2, device="cuda")
        loss_r = ref.loss(x, y, keep_prob=0.8, masks=masks)
        loss_r.backward()
        assert abs(loss_n - float(loss_r.detach())) < 5e-6
        for k, p in ref.trainable().items():
            g_ref = p.grad.cpu().numpy()
            assert np.abs(grads_n[k] - g_ref).max() <= 2e-3 * np.abs(g_ref).max() + 1e-9, k
        # explicit masks replayed through the native step (the mask-tensor path) give the same result
        loss_m, grads_m = tr.gradients(x, y, masks=masks)
        assert abs(loss_m - loss_n) < 1e-6
        assert all(np.allclose(grads_m[k], grads_n[k], rtol=1e-5, atol=1e-9) for k in grads_n)
        # a training step advances the optimizer's step counter: new masks
        tr.train_step(x, y)
        masks2 = tr.step_impl.dropout_scales(64)
        assert not np.array_equal(masks2[(0, "fw")], masks[(0, "fw")])
        # the graph-captured step draws new masks on every replay as well (the kernels read the counter from device memory)
        tg = Trainer(ckpt_weights, 3, 2, "RMSProp", 1e-3, keep_prob=0.8, device="cuda", native=True, seed=5, use_graph=True)
        l1 = [tg.train_step(x, y) for _ in range(3)]
        te = Trainer(ckpt_weights, 3, 2, "RMSProp", 1e-3, keep_prob=0.8, device="cuda", native=True, seed=5, use_graph=False)
        l2 = [te.train_step(x, y) for _ in range(3)]
        assert np.allclose(l1, l2, atol=1e-6), (l1, l2)                    # same seed: graph replay == eager, step by step
        tg.engine.close(); te.engine.close()
    finally:
        tr.engine.close()


def test_cli_end_to_end(tmp_path, ckpt_weights, monkeypatch):
    """`catfish -i IN -s OUT -c 300` (catfish/catfish:18-94): model directory with ResNetRNN.txt + a TF
    checkpoint-V2 bundle (written here from the exported tensors), a directory of reads, chunk coordinates out."""
    import json
    from click.testing import CliRunner
    from catfish_amd import checkpoint, cli
    net = tmp_path / "ResNetRNN"
    (net / "checkpoints").mkdir(parents=True)
    (net / "ResNetRNN.txt").write_text("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\n"
                                       "learning_rate: 0.001\nlayer_size: 64\nn_layers: 3\nkeep_prob: 0.8\n"
                                       "layer_size_res: 32\nn_layers_res: 2\n")
    checkpoint.write_checkpoint(str(net / "checkpoints" / "ckpnt-30000"), ckpt_weights)
    reads = tmp_path / "reads"
    reads.mkdir()
    dacs = {}
    for i, n in enumerate((4096, 2500, 700)):
        d = oracle.synthetic_dac(1, n, seed=500 + i)[0]
        np.save(reads / ("read%d.npy" % i), d)
        dacs["read%d.npy" % i] = d
    monkeypatch.chdir(tmp_path)                      # the reference resolves "ResNetRNN" relative to the CWD (:44)
    res = CliRunner().invoke(cli._build_click_main(), ["-i", str(reads), "-s", str(tmp_path / "out"), "-c", "300"])
    assert res.exit_code == 0, res.output
    hp = json.load(open(tmp_path / "out" / "TEMP" / "hp_positions.json"))
    nonhp = json.load(open(tmp_path / "out" / "TEMP" / "nonhp_positions.json"))
    for name, d in dacs.items():
        spans, length, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        if spans:
            merged = cli.merge_positions([list(s) for s in spans], length, 300)
            assert hp[name] == merged
            assert nonhp[name] == cli.nonhp_complement(merged, length)
        else:
            assert name not in hp
    # the split step (catfish/catfish:85-92 -> split_f5.py:8-81): signal[s0:s1] of every chunk as an int16 .npy, HP chunks first,
    # the index running on into the non-HP ones; reads without homopolymers are not split
    want = {}
    for name in hp:
        for k, (s0, s1) in enumerate(hp[name] + nonhp[name]):
            want["%s/%s_%d.npy" % ("HP" if k < len(hp[name]) else "nonHP", name.split(".")[0], k)] = dacs[name][s0:s1]
    assert hp and want
    got = {"%s/%s" % (d, f) for d in ("HP", "nonHP") for f in os.listdir(tmp_path / "out" / "TEMP" / d)}
    assert got == set(want)
    for rel, samples in want.items():
        back = np.load(tmp_path / "out" / "TEMP" / rel)
        assert back.dtype == np.int16 and np.array_equal(back, samples)
    assert "Finished splitting the raw signals in" in res.output
    # second run into the same directory fails like the reference (os.makedirs on an existing TEMP/HP)
    res2 = CliRunner().invoke(cli._build_click_main(), ["-i", str(reads), "-s", str(tmp_path / "out")])
    assert res2.exit_code != 0


def _write_model_dir(tmp_path, ckpt_weights):
    from catfish_amd import checkpoint
    net = tmp_path / "ResNetRNN"
    (net / "checkpoints").mkdir(parents=True)
    (net / "ResNetRNN.txt").write_text("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\n"
                                       "learning_rate: 0.001\nlayer_size: 64\nn_layers: 3\nkeep_prob: 0.8\n"
                                       "layer_size_res: 32\nn_layers_res: 2\n")
    checkpoint.write_checkpoint(str(net / "checkpoints" / "ckpnt-30000"), ckpt_weights)
    return net


def test_sharded_runner_world1_real_engine(model, ckpt_weights, tmp_path):
    """BASELINE configs[2] plumbing at world_size 1: the sharded runner (catfish/catfish:50-56 partitioned) drives the
    REAL HIP engine through the streaming pipeline and equals the per-read oracle; float inputs take the host
    normalisation branch."""
    from catfish_amd import sharding
    lens = [4096, 36, 700, 35, 140, 999, 70, 512, 64, 300, 2000, 3333, 37, 4096]
    dacs = [oracle.synthetic_dac(1, max(n, 2), seed=900 + i)[0][:n] for i, n in enumerate(lens)]
    want = []
    for d in dacs:
        spans, n, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        want.append((spans, n))
    got = sharding.infer_reads_sharded(model, dacs, max_samples_per_batch=6000)          # several batches, one oversize read
    assert got == want
    paths = []
    for i, d in enumerate(dacs):
        p = tmp_path / ("r%02d.npy" % i)
        np.save(p, d)
        paths.append(str(p))
    assert sharding.infer_files_sharded(model, paths, max_samples_per_batch=6000) == want
    # explicit rank of a 3-rank job without a process group: only that shard is touched, gather is refused
    with pytest.raises(RuntimeError):
        sharding.infer_reads_sharded(model, dacs, rank=1, world_size=3)
    # float (already calibrated) traces: host normalisation branch
    flt = [d.astype(np.float64) * 0.25 + 3.0 for d in dacs[:4]]
    got_f = sharding.infer_reads_sharded(model, flt)
    for (spans, n), f in zip(got_f, flt):
        w_spans, w_n, _ = oracle.infer_read(oracle.normalize_raw_signal(f), ckpt_weights, np.float32)
        assert (spans, n) == (w_spans, w_n)


def test_files_go_through_the_native_loader_and_fall_back_per_batch(model, ckpt_weights, tmp_path, monkeypatch):
    """sharding.infer_files_sharded / chunk_files_sharded on the real engine: int16 .npy reads are read by the library's
    thread pool straight into the pipeline's pinned staging buffer (ReadPipeline.submit_files); a batch holding another
    format (.npz, float .npy) or more samples than its size on disk suggested goes through infer.load_dac -- same results
    either way, in the order of the file names."""
    from catfish_amd import chunks, cli, sharding
    from catfish_amd.pipeline import ReadPipeline
    lens = [4096, 36, 700, 35, 140, 999, 70, 512, 64, 300, 2000, 3333, 37, 4096, 9000]
    dacs = [oracle.synthetic_dac(1, max(n, 2), seed=400 + i)[0][:n] for i, n in enumerate(lens)]
    paths = []
    for i, d in enumerate(dacs):
        if i == 5:
            p = tmp_path / ("r%02d.npz" % i)
            np.savez(p, raw=d)                                   # the reference's NPZ layout: general loader
        elif i == 9:
            p = tmp_path / ("r%02d.npy" % i)
            np.save(p, d.astype(np.int64))                       # integer codes in another dtype: general loader
        else:
            p = tmp_path / ("r%02d.npy" % i)
            np.save(p, d)
        paths.append(str(p))
    want = []
    for d in dacs:
        spans, n, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        want.append((spans, n))
    calls = {"native": 0, "fallback": 0}
    real = ReadPipeline.submit_files

    def counted(self, batch, n_threads=4):
        t = real(self, batch, n_threads)
        calls["native" if t is not None else "fallback"] += 1
        return t

    monkeypatch.setattr(ReadPipeline, "submit_files", counted)
    got = sharding.infer_files_sharded(model, paths, max_samples_per_batch=6000)      # several batches, one oversize read
    assert got == want
    assert calls["native"] >= 3 and calls["fallback"] >= 2                            # both kinds of batch occurred
    table = sharding.chunk_files_sharded(model, paths, chunk_size=300, max_samples_per_batch=6000)
    hp, nonhp = table.to_dicts(paths)
    for p, (spans, n) in zip(paths, want):
        merged, non = cli.chunks_of_read([list(s) for s in spans], n, 300)
        assert hp.get(p) == merged and json.dumps(nonhp[p]) == json.dumps(non)
    assert isinstance(table, chunks.ChunkTable) and list(table.lengths) == lens


@pytest.mark.timeout(600)
def test_cli_two_ranks_share_one_gpu(tmp_path, ckpt_weights):
    """The multi-GPU entry point end to end: `catfish --gpus 2`, which starts `torch.distributed.run --nproc-per-node 2
    -m catfish_amd.cli` as a child process; both ranks pinned to cuda:0 by CATFISH_DEVICE because this box has one GPU.
    Rank 0 gathers over gloo and writes the chunk coordinates; they must equal the per-read oracle."""
    import json
    import subprocess
    import sys
    from catfish_amd import cli
    from conftest import ROOT
    _write_model_dir(tmp_path, ckpt_weights)
    reads = tmp_path / "reads"
    reads.mkdir()
    dacs = {}
    for i, n in enumerate((4096, 2500, 700, 36, 3000, 1500, 5000)):
        d = oracle.synthetic_dac(1, n, seed=700 + i)[0]
        np.save(reads / ("read%d.npy" % i), d)
        dacs["read%d.npy" % i] = d
    env = dict(os.environ, CATFISH_DEVICE="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    # `catfish --gpus 2`: the CLI starts torch.distributed.run (--nproc-per-node 2 -m catfish_amd.cli ...) as a child process
    env.pop("MASTER_PORT", None)                      # the launcher picks a free rendezvous port
    cmd = [sys.executable, "-m", "catfish_amd.cli", "-i", str(reads), "-s", str(tmp_path / "out"), "-c", "300", "--gpus", "2"]
    res = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=500)
    assert res.returncode == 0, res.stdout[-3000:]
    hp = json.load(open(tmp_path / "out" / "TEMP" / "hp_positions.json"))
    nonhp = json.load(open(tmp_path / "out" / "TEMP" / "nonhp_positions.json"))
    assert set(nonhp) == set(dacs)
    for name, d in dacs.items():
        spans, length, _ = oracle.infer_read(oracle.normalize_raw_signal(d), ckpt_weights, np.float32)
        merged, non = cli.chunks_of_read([list(s) for s in spans], length, 300)
        if merged is not None:
            assert hp[name] == merged
            assert nonhp[name] == non
        else:
            assert name not in hp


@pytest.mark.timeout(300)
def test_rccl_probe_of_the_bench_succeeds_with_one_rank(tmp_path):
    """bench.py brings RCCL ("nccl" on ROCm) up as a PROBE beside its host group (`bench.probe_rccl`).  Two ranks on this box's ONE card
    cannot succeed ("Duplicate GPU detected" -- the rehearsal lines record it); ONE rank can: a second process group over RCCL, one
    all-reduce of a device scalar, blocking wait, destroyed again -- the success path of the probe (`ok`, `ranks_seen` = world) and
    proof that this image's RCCL loads and runs a collective on the card.  In a child process: a process group is per process."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "import bench\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "res = bench.probe_rccl(dist, torch, 0, timeout_s=120)\n"
        "one = torch.ones(1); dist.all_reduce(one)\n"
        "dist.destroy_process_group()\n"
        "print('PROBE ' + json.dumps(res))\n" % ROOT)
    from catfish_amd.cli import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True,
                         env=env, timeout=250, cwd=str(tmp_path))
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("PROBE ")]
    assert run.returncode == 0 and lines, run.stdout[-3000:]
    res = json.loads(lines[-1][len("PROBE "):])
    assert res["seconds"] < 150 and res["timeout_s"] == 120, res            # bounded either way
    if not res["ok"]:
        # whether RCCL comes up is a property of the box, not of the product (the data path has no collective): the probe must then
        # have SAID so -- that is its job -- and this test has nothing to show
        assert res["error"], res
        pytest.skip("RCCL did not come up on this box: %s" % res["error"])
    assert res["ranks_seen"] == 1 and res["error"] is None, res


@pytest.mark.timeout(600)
def test_configs2_shard_at_full_per_rank_size(model, ckpt_weights):
    """BASELINE configs[2] at ONE rank's full size -- 12 500 synthetic 4096-sample reads (seed 1, the reads `bench.py`'s sharded legs
    use: 51 M samples, 700 x what the oracle classifies per second and core) -- through the sharded runner on the real engine.
    Size-independent properties: (1) windows are independent, so how many reads share a launch must not matter: batches of 256 and
    of 1100 reads give the same flat span table, bit for bit; (2) the blocks the 8-rank job would cut (`shard_contiguous`) tile the
    list and a block classified on its own equals its rows of the whole; (3) every read has its true length and its spans lie
    inside [-11, length + 16] in ascending order; plus an oracle sample of six reads (first / last of the list and of two blocks)."""
    import bench
    from catfish_amd import sharding
    n = 12500
    reads = [bench.squiggle_dac(np.random.default_rng([1, i]), bench.READ_LEN) for i in range(n)]
    lengths = [bench.READ_LEN] * n
    whole = sharding.infer_reads_sharded(model, reads, lengths=lengths, max_samples_per_batch=256 * bench.READ_LEN, as_table=True)
    other = sharding.infer_reads_sharded(model, reads, lengths=lengths, max_samples_per_batch=1100 * bench.READ_LEN, as_table=True)
    assert len(whole) == n and bool((whole.lengths == bench.READ_LEN).all())
    for name in ("read_of", "start", "end", "lengths"):
        assert np.array_equal(getattr(whole, name), getattr(other, name)), name
    assert np.all(np.diff(whole.read_of) >= 0) and np.all(whole.start >= -11) and np.all(whole.end <= bench.READ_LEN + 16)
    assert np.all(whole.end - whole.start >= 15 + 27)                                    # correct_short's 15 samples + the 11 / 16 extension
    same_read = np.diff(whole.read_of) == 0
    assert np.all(whole.start[1:][same_read] > whole.end[:-1][same_read] - 27)           # runs of one read ascend (extensions may overlap)
    shards = sharding.shard_contiguous([sharding.windows_of(x) for x in lengths], 8)
    assert [i for blk in shards for i in blk] == list(range(n)) and all(len(b) in (1562, 1563) for b in shards)
    for r in (0, 7):
        blk = shards[r]
        alone = sharding.infer_reads_sharded(model, [reads[i] for i in blk], max_samples_per_batch=256 * bench.READ_LEN, as_table=True)
        rows = (whole.read_of >= blk[0]) & (whole.read_of <= blk[-1])
        assert np.array_equal(alone.read_of + blk[0], whole.read_of[rows])
        assert np.array_equal(alone.start, whole.start[rows]) and np.array_equal(alone.end, whole.end[rows])
    for i in (0, n - 1, shards[3][0], shards[3][-1], shards[7][0], 6250):
        spans, length, _ = oracle.infer_read(oracle.normalize_raw_signal(reads[i]), ckpt_weights, np.float32)
        assert whole.read(i) == (spans, length), i
    assert len(whole.start) > n                                                          # the squiggles do hold homopolymer calls


def test_configs1_at_full_size_is_invariant_to_the_launch_size(ckpt_weights):
    """BASELINE configs[1] at its full size -- 10 000 synthetic 4096-sample reads, fp32 -- which the oracle cannot follow
    (it classifies ~75 k samples/s per core; this is 41 M): windows are independent, so how many reads share a launch must
    not matter.  The benchmark's 256-read launches (per-layer biGRU launches, four chip rounds each) against 1024-read
    launches (120 832 windows: the dynamically scheduled single launch of all three layers) give bit-identical
    probabilities and identical spans for every read; lengths come back exact; three reads are checked against the fp64
    oracle at the 1e-4 gate; and device normalisation + post-processing make the spans equal the per-read oracle pipeline."""
    from catfish_amd import batching
    from catfish_amd.engine import HipEngine
    n_reads, length = 10000, 4096
    dacs = list(oracle.synthetic_dac(n_reads, length, seed=0))
    eng = HipEngine(ckpt_weights, device=0, max_windows_per_pass=1024 * 118)
    try:
        small, p_small = batching.infer_reads_dac(eng, dacs, max_windows=256 * 118, return_probs=True)
        digest_small = [float(p.sum(dtype=np.float64)) for p in p_small]
        keep = {i: p_small[i].copy() for i in (0, 4999, 9999)}
        del p_small
        big, p_big = batching.infer_reads_dac(eng, dacs, max_windows=1024 * 118, return_probs=True)
        assert [n for _s, n in small] == [length] * n_reads == [n for _s, n in big]
        assert small == big
        assert digest_small == [float(p.sum(dtype=np.float64)) for p in p_big]
        for i, p in keep.items():
            assert np.array_equal(p, p_big[i])
            x, _pad = oracle.pad_and_window(oracle.normalize_raw_signal(dacs[i]))
            want = oracle.forward(x, ckpt_weights, np.float64)[:length]
            assert np.abs(p - want).max() < 1e-4
            lab = oracle.correct_short(oracle.class_from_threshold(oracle.forward(x, ckpt_weights, np.float32)[:length]))
            assert big[i][0] == (oracle.hp_in_pred(lab) if lab.any() else [])
        eng.check_error()
    finally:
        eng.close()


def _normalise_on_device(engine, reads):
    import torch
    from catfish_amd import infer
    dev = torch.device("cuda", 0)
    lens = [len(r) for r in reads]
    dac_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    n_win = [(n + infer.padding_size_for(n)) // 35 for n in lens]
    win_off = np.concatenate(([0], np.cumsum(n_win))).astype(np.int64)
    x = engine.normalize_device(torch.from_numpy(np.concatenate(reads)).to(dev), torch.from_numpy(dac_off).to(dev),
                                torch.from_numpy(win_off).to(dev)).cpu().numpy()
    return x, win_off


def test_register_ingest_kernel_is_bit_exact_at_every_length_class(model, monkeypatch):
    """normalize_regs_kernel<16> (reads <= 4096 samples, in registers), <64> (<= 16 384) and the radix path behind it (longer), against
    numpy's float64 normalisation cast to float32 (catfish/infer.py:96-105) AND against round 1's kernel (CATFISH_INGEST_V1 behind
    the debug switch): every length class and their edges, odd and even lengths (the upper middle element comes from a different
    code path), squiggles (a few hundred distinct codes: heavy ties), the full int16 range, constant reads (MAD = 0: inf / nan like
    numpy), two-valued reads, reads next to each other whose classes differ."""
    from catfish_amd import infer
    rng = np.random.default_rng(11)
    lens = [1, 2, 3, 4, 255, 256, 257, 511, 512, 1000, 4095, 4096, 4097, 5000, 8191, 8192, 16383, 16384, 16385, 20001, 40000]
    reads = [np.clip(np.rint(rng.normal(500, 60, size=n)), 0, 2047).astype(np.int16) for n in lens]
    reads += [rng.integers(-32768, 32768, size=n).astype(np.int16) for n in (2, 77, 4096, 9999, 16384, 30000)]
    reads += [np.full(n, 417, dtype=np.int16) for n in (1, 2, 4096, 5000)]                       # constant: scale 0
    reads += [np.array([-32768, 32767] * 50, dtype=np.int16), np.array([-32768] * 7 + [32767] * 8, dtype=np.int16),
              np.array([32767, -32768, 0], dtype=np.int16), np.repeat(np.array([3, 9], dtype=np.int16), 2048)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    new, win_off = _normalise_on_device(model.engine, reads)
    for i, r in enumerate(reads):
        with np.errstate(divide="ignore", invalid="ignore"):
            want = infer.normalize_raw_signal(r, "median").astype(np.float32)
        got = new[win_off[i]:win_off[i + 1]].reshape(-1)
        assert np.array_equal(got[:len(r)], want, equal_nan=True), (i, len(r))
        assert np.all(got[len(r):] == 0), (i, len(r))
    monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
    monkeypatch.setenv("CATFISH_INGEST_V1", "1")
    old, _ = _normalise_on_device(model.engine, reads)
    assert np.array_equal(old.view(np.uint32), new.view(np.uint32))                                # bit for bit, nan payloads included


def test_bitmask_postprocess_kernel_matches_oracle_and_round_1_kernel(model, monkeypatch):
    """postprocess_bits_kernel against the oracle's threshold + correct_short (catfish/infer.py:128-138, 174-198) per read and against
    round 1's per-sample kernel bit for bit: many short reads (several per 64-sample word), runs of exactly min_run - 1 / min_run /
    min_run + 1, runs that touch the end of a read and padding filled with ones right behind it, runs across word and chunk
    (62-word) boundaries, a total that is not a multiple of 64, min_run 1 .. 64 and 65 (the fallback), other thresholds."""
    torch = pytest.importorskip("torch")
    from catfish_amd import batching
    rng = np.random.default_rng(5)
    lens = [1, 2, 14, 15, 16, 29, 30, 31, 35, 36, 63, 64, 65, 70, 105, 127, 128, 129, 700, 3967, 3968, 3969, 4096, 8000, 12345] + \
        rng.integers(1, 200, size=60).tolist()
    sigs = [np.zeros(n) for n in lens]
    pk = batching.pack_reads(sigs)
    total = pk.n_windows * 35
    dev = torch.device("cuda", 0)
    offs, lengths = torch.from_numpy(pk.sample_offsets).to(dev), torch.from_numpy(pk.lengths).to(dev)
    for min_run, threshold in ((15, 0.5), (1, 0.5), (2, 0.5), (16, 0.25), (33, 0.5), (64, 0.5), (65, 0.5), (14, 0.75)):
        probs = np.ones(total, dtype=np.float32)                      # padding stays 1.0: it must still come out as 0
        per_read = []
        for i, n in enumerate(lens):
            # runs of controlled lengths around min_run, separated by single zeros or longer gaps
            p = np.zeros(n, dtype=np.float32)
            pos = int(rng.integers(0, 3))
            while pos < n:
                run = int(rng.choice([1, min_run - 1, min_run, min_run + 1, 2 * min_run + 3, int(rng.integers(1, 90))]))
                run = max(1, run)
                p[pos:pos + run] = rng.uniform(threshold, 1.0, size=len(p[pos:pos + run]))
                pos += run + int(rng.choice([1, 1, 2, 20]))
            p[p == 0] = rng.uniform(0.0, threshold * 0.999, size=int((p == 0).sum()))
            if i % 5 == 0:
                p[-min(n, min_run + 2):] = 1.0                            # a run that ends with the read, ones in the padding behind it
            per_read.append(p)
            probs[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = p
        monkeypatch.delenv("CATFISH_INGEST_V1", raising=False)
        new = model.engine.postprocess_device(torch.from_numpy(probs).to(dev), offs, lengths, threshold=threshold, min_run=min_run).cpu().numpy()
        for i, n in enumerate(lens):
            want = oracle.correct_short(oracle.class_from_threshold(per_read[i], threshold), min_run)
            got = new[pk.sample_offsets[i]:pk.sample_offsets[i] + n]
            assert np.array_equal(got, want), (min_run, i, n)
            assert not new[pk.sample_offsets[i] + n:pk.sample_offsets[i + 1]].any(), (min_run, i)
        monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
        monkeypatch.setenv("CATFISH_INGEST_V1", "1")
        old = model.engine.postprocess_device(torch.from_numpy(probs).to(dev), offs, lengths, threshold=threshold, min_run=min_run).cpu().numpy()
        assert np.array_equal(old, new), min_run
    # a labels buffer that does not start on a 16-byte boundary (a view into a larger allocation): the C ABI takes any device pointer
    monkeypatch.delenv("CATFISH_INGEST_V1", raising=False)
    big = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
    for shift in (1, 3, 8):
        view = big[shift:shift + total]
        got = model.engine.postprocess_device(torch.from_numpy(probs).to(dev), offs, lengths, threshold=threshold, min_run=min_run, out=view)
        assert np.array_equal(got.cpu().numpy(), new), shift
    # a total that is not a multiple of 64 (the last word goes out byte by byte), all positive, one read
    monkeypatch.delenv("CATFISH_INGEST_V1", raising=False)
    for n in (1, 63, 65, 100, 64 * 62 + 7):
        p = torch.ones(n, device=dev)
        o = torch.tensor([0, n], dtype=torch.int64, device=dev)
        ln = torch.tensor([n - (n > 20) * 3], dtype=torch.int64, device=dev)
        got = model.engine.postprocess_device(p, o, ln).cpu().numpy()
        real = int(ln.item())
        assert got[:real].tolist() == ([1] * real if real >= 15 else [0] * real) and not got[real:].any(), n


def test_postprocess_keeps_runs_inside_their_read_when_reads_are_packed_without_padding(model, monkeypatch):
    """the bit-mask kernel eroded across one flat validity mask, so with a read table WITHOUT padding
    (read_lengths[r] == read_offsets[r + 1] - read_offsets[r]: legal through the C ABI, never produced by the framework's packers) a
    7-run ending read r and an 8-run starting read r + 1 survived correct_short as one 15-run, and the spans variant emitted one span
    across the boundary.  Runs are cut where a read begins now: labels equal the per-read oracle (infer.py:174-198) and round 1's
    per-sample kernel bit for bit, spans equal the per-read runs, on both code paths, with boundaries at every position of a word."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    lens = [7, 8, 64, 1, 14, 15, 16, 63, 65, 100, 30, 30, 128, 5, 9, 4000, 33] + rng.integers(1, 90, size=80).tolist()
    offs_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offs_h[-1])
    dev = torch.device("cuda", 0)
    offs, lengths = torch.from_numpy(offs_h).to(dev), torch.from_numpy(np.asarray(lens, np.int64)).to(dev)
    for min_run in (15, 2, 33, 64, 65):
        per_read = []
        for i, n in enumerate(lens):
            p = (rng.random(n) < 0.35).astype(np.float32)
            if i % 2 == 0:
                p[-min(n, max(1, min_run // 2)):] = 1.0                    # a short run ending with the read ...
            else:
                p[:min(n, min_run - min_run // 2)] = 1.0                   # ... meets one starting the next: min_run together
            if i % 5 == 0:
                p[:] = 1.0                                                 # whole reads positive: long runs on both sides of a boundary
            per_read.append(p)
        probs = torch.from_numpy(np.concatenate(per_read)).to(dev)
        want = np.concatenate([np.asarray(oracle.correct_short(oracle.class_from_threshold(p), min_run), dtype=np.uint8) for p in per_read])
        want_runs = []
        for i, p in enumerate(per_read):
            lab = np.concatenate([[0], np.asarray(oracle.correct_short(oracle.class_from_threshold(p), min_run)), [0]])
            edges = np.flatnonzero(np.diff(lab))
            want_runs += [(int(offs_h[i] + a), int(offs_h[i] + b)) for a, b in zip(edges[0::2], edges[1::2])]
        assert any(a == o for a, _b in want_runs for o in offs_h[1:-1]) and any(b == o for _a, b in want_runs for o in offs_h[1:-1])
        for v1 in (False, True):
            if v1:
                monkeypatch.setenv("CATFISH_DEBUG_KNOBS", "1")
                monkeypatch.setenv("CATFISH_INGEST_V1", "1")
            else:
                monkeypatch.delenv("CATFISH_INGEST_V1", raising=False)
            got = model.engine.postprocess_device(probs, offs, lengths, min_run=min_run).cpu().numpy()
            assert np.array_equal(got, want), (min_run, v1)
            for with_labels in (True, False):                              # labels NULL: also on the fallback path
                res = model.engine.postprocess_spans_device(probs, offs, lengths, min_run=min_run, labels=with_labels)
                assert list(zip(res[0].tolist(), res[1].tolist())) == want_runs, (min_run, v1, with_labels)
                if with_labels:
                    assert np.array_equal(res[2].cpu().numpy(), want), (min_run, v1)
    monkeypatch.delenv("CATFISH_INGEST_V1", raising=False)


def test_cli_directory_of_mixed_formats_and_many_batches(tmp_path, ckpt_weights):
    """`cli.run_pipeline` over a directory the native listing path cannot take whole: int16 ``.npy`` vectors (the fast path: listing
    ranges, files preloaded by the helper thread), an ``.npz`` (key ``raw``), a headerless ``.bin``, an ``.npy`` holding int32 codes
    and one holding an already NORMALISED float trace -- batches that contain one of those fall back to ``infer.load_dac`` (and the
    float read to the host normalisation branch) while their neighbours stay on the fast path; a small pass capacity forces many
    batches, ramped first.  Every read must equal the per-read oracle, in the listing's (bytewise) order (catfish/catfish:50-82)."""
    import contextlib
    import io
    import json
    from catfish_amd import cli, sharding
    net = _write_model_dir(tmp_path, ckpt_weights)
    reads = tmp_path / "reads"
    reads.mkdir()
    raw = {}
    for i in range(34):
        d = oracle.synthetic_dac(1, 600 + 97 * (i % 9), seed=700 + i)[0]
        name = "read_%02d.npy" % i
        if i == 5:
            name = "read_05.npz"
            np.savez(reads / name, raw=d)
        elif i == 11:
            name = "read_11.bin"
            d.astype("<i2").tofile(reads / name)
        elif i == 20:
            np.save(reads / name, d.astype(np.int32))                          # integer codes of another width: still a DAC read
        elif i == 27:
            np.save(reads / name, oracle.normalize_raw_signal(d))               # float64: normalised once more on the host, like load_raw would
        else:
            np.save(reads / name, d)
        raw[name] = d
    orig = sharding.RAMP
    timings = {}
    with contextlib.redirect_stdout(io.StringIO()):
        res = cli.run_pipeline(str(reads), str(tmp_path / "out"), chunk_size=300, network_path=str(net), device=0, timings=timings,
                               gather_table=True)
    assert sharding.RAMP == orig and res["reads"] == 34 and res["files"] == sorted(raw)
    hp = json.load(open(tmp_path / "out" / "TEMP" / "hp_positions.json"))
    nonhp = json.load(open(tmp_path / "out" / "TEMP" / "nonhp_positions.json"))
    assert list(nonhp) == sorted(raw)                                           # one rank: the documents follow the listing's order
    for name, d in raw.items():
        sig = oracle.normalize_raw_signal(d)
        if name == "read_27.npy":
            sig = oracle.normalize_raw_signal(sig)
        spans, length, _ = oracle.infer_read(sig, ckpt_weights, np.float32)
        merged, non = cli.chunks_of_read([list(s) for s in spans], length, 300)
        assert hp.get(name) == merged and nonhp[name] == json.loads(json.dumps(non)), name
    # the same directory in batches of a few reads each (ramped 1/8, 3/8, then full): same documents
    from catfish_amd import neural_network
    model = neural_network.load_network("ResNetRNN", str(net), checkpoint=30000, device=0, max_windows_per_pass=256)
    listing, sizes = sharding.shared_listing(str(reads))
    mine, table = sharding.chunk_files_local(model, sharding.ListingPaths(listing), 300, max_samples_per_batch=256 * 35, rank=0, world_size=1,
                                             file_sizes=sizes)
    assert mine == list(range(34))
    hp2, non2 = table.to_dicts(listing.names())
    assert json.loads(json.dumps(hp2)) == hp and json.loads(json.dumps(non2)) == nonhp


def test_file_driven_shard_at_scale_is_invariant_to_batching(tmp_path, model):
    """The file-driven path behind the CLI at a size the oracle cannot follow: 1500 files of 1 .. 40 000 samples (log-uniform: reads
    below a window, reads past the ingest kernel's register classes at 4096 and 16 384, a few beyond the staging buffer's estimate),
    through ``chunk_files_local`` over a native listing -- listing ranges, ramped batches cut by padded samples, the files of batch
    k + 1 preloaded by the helper thread, oversize batches re-cut -- against ``batching.infer_reads_dac`` on the same reads loaded in
    Python and packed by length buckets.  Windows are independent, so the spans of every read must be IDENTICAL whichever way the
    reads were grouped, staged and launched; a read dropped, duplicated or shifted by any of that machinery shows here."""
    from catfish_amd import batching, chunks, infer, sharding
    rng = np.random.default_rng(17)
    lens = np.rint(np.exp(rng.uniform(0, np.log(40000), size=1500))).astype(int)
    lens[:6] = [1, 34, 35, 36, 4096, 16384]
    d = tmp_path / "reads"
    d.mkdir()
    for i, n in enumerate(lens):
        np.save(d / ("r%05d.npy" % ((i * 7919) % 100000)), oracle.synthetic_dac(1, max(int(n), 2), seed=3000 + i)[0][:int(n)])
    listing, sizes = sharding.shared_listing(str(d))
    names = listing.names()
    assert len(names) == 1500
    mine, table = sharding.chunk_files_local(model, sharding.ListingPaths(listing), 1000, max_samples_per_batch=8192 * 35, rank=0,
                                             world_size=1, file_sizes=sizes)
    assert mine == list(range(1500)) and len(table) == 1500
    reads = [infer.load_dac(str(d / n)) for n in names]
    want = batching.infer_reads_dac(model, reads, max_windows=4096)
    assert [int(x) for x in table.lengths] == [len(r) for r in reads] == [w[1] for w in want]
    bounds = np.concatenate(([0], np.cumsum([len(w[0]) for w in want])))
    flat = np.array([s for w in want for s in w[0]], dtype=np.int64).reshape(-1, 2)
    ref = chunks.ChunkTable.from_spans(bounds, flat[:, 0], flat[:, 1], [w[1] for w in want], 1000)
    for a, b in ((table.hp_bounds, ref.hp_bounds), (table.hp_start, ref.hp_start), (table.hp_end, ref.hp_end),
                 (table.nonhp_bounds, ref.nonhp_bounds), (table.nonhp_start, ref.nonhp_start), (table.nonhp_end, ref.nonhp_end)):
        assert np.array_equal(a, b)
    assert int(table.has_hp.sum()) > 100                                                   # (the comparison is not vacuous)


def test_fused_postprocess_spans_equals_the_two_calls_it_replaces(model):
    """cf_postprocess_spans (threshold + correct_short + run boundaries in one launch) against cf_postprocess followed
    by cf_spans on the same probabilities: the same sorted starts / ends and, when asked for, the same labels -- many short reads,
    runs around min_run, runs touching read ends with ones in the padding behind them, runs across word and 62-word chunk
    boundaries, a total that is not a multiple of 64; min_run 1, 15, 64 and 65 (the two-kernel fallback); a max_runs too small for
    the runs found (counts say so, the lists are truncated, the wrapper retries)."""
    torch = pytest.importorskip("torch")
    from catfish_amd import batching
    rng = np.random.default_rng(8)
    lens = [1, 15, 16, 35, 36, 64, 129, 700, 3967, 3968, 3969, 4096, 9000] + rng.integers(1, 300, size=80).tolist()
    pk = batching.pack_reads([np.zeros(n) for n in lens])
    total = pk.n_windows * 35
    dev = torch.device("cuda", 0)
    offs, lengths = torch.from_numpy(pk.sample_offsets).to(dev), torch.from_numpy(pk.lengths).to(dev)
    eng = model.engine
    for min_run in (15, 1, 64, 65):
        probs = np.ones(total, dtype=np.float32)
        for i, n in enumerate(lens):
            p = np.zeros(n, dtype=np.float32)
            pos = int(rng.integers(0, 3))
            while pos < n:
                run = max(1, int(rng.choice([1, min_run - 1, min_run, min_run + 1, 3 * min_run, int(rng.integers(1, 120))])))
                p[pos:pos + run] = 0.9
                pos += run + int(rng.choice([1, 1, 3, 40]))
            p[p == 0] = 0.1
            if i % 4 == 0:
                p[-min(n, min_run + 1):] = 0.9
            probs[pk.sample_offsets[i]:pk.sample_offsets[i] + n] = p
        d_probs = torch.from_numpy(probs).to(dev)
        labels = eng.postprocess_device(d_probs, offs, lengths, min_run=min_run)
        want_s, want_e = eng.spans_device(labels)
        got_s, got_e, got_labels = eng.postprocess_spans_device(d_probs, offs, lengths, min_run=min_run, labels=True)
        assert np.array_equal(got_s, want_s) and np.array_equal(got_e, want_e), min_run
        assert torch.equal(got_labels, labels) and len(want_s) > 50, min_run
        only_s, only_e = eng.postprocess_spans_device(d_probs, offs, lengths, min_run=min_run)           # labels never written
        assert np.array_equal(only_s, want_s) and np.array_equal(only_e, want_e)
        small_s, small_e = eng.postprocess_spans_device(d_probs, offs, lengths, min_run=min_run, max_runs=7)   # too small: the wrapper retries
        assert np.array_equal(small_s, want_s) and np.array_equal(small_e, want_e)
        # and against the per-read reference rules
        spans = batching.spans_from_runs(got_s, got_e, pk.sample_offsets, pk.n_reads) if hasattr(batching, "spans_from_runs") else None
        if spans is not None and min_run == 15:
            for i, n in enumerate(lens):
                want = oracle.correct_short(oracle.class_from_threshold(probs[pk.sample_offsets[i]:pk.sample_offsets[i] + n]))
                assert spans[i] == (oracle.hp_in_pred(want) if np.asarray(want).any() else []), i
    empty_s, empty_e = eng.postprocess_spans_device(torch.zeros(640, device=dev), torch.tensor([0, 640], device=dev),
                                                    torch.tensor([630], device=dev))
    assert len(empty_s) == 0 and len(empty_e) == 0


def test_ingest_and_postprocess_kernels_at_scale_random(model):
    """The round-5 ingest and post-processing kernels on 1500 reads of 1 .. 20 000 samples at once (every register class and the
    radix path in one launch, reads of all classes next to each other): normalisation bit-exact against numpy per read; corrected labels
    and run boundaries against the vectorised host rules (``infer.correct_short``, themselves pinned by reference-executed goldens),
    with smooth random scores (runs of every length) and padding filled with ones."""
    torch = pytest.importorskip("torch")
    from catfish_amd import batching, infer
    rng = np.random.default_rng(23)
    lens = np.rint(np.exp(rng.uniform(0, np.log(20000), size=1500))).astype(int)
    reads = []
    for i, n in enumerate(lens):
        kind = i % 4
        if kind == 0:
            r = np.clip(np.rint(rng.normal(500, 60, size=n)), 0, 2047)
        elif kind == 1:
            r = rng.integers(-32768, 32768, size=n)
        elif kind == 2:
            r = np.repeat(np.clip(np.rint(rng.normal(450, 80, size=n // 9 + 1)), 0, 2047), 9)[:n]      # dwell-like plateaus: heavy ties
        else:
            r = rng.integers(400, 404, size=n)                                                          # four codes only
        reads.append(r.astype(np.int16))
    x, win_off = _normalise_on_device(model.engine, reads)
    for i, r in enumerate(reads):
        with np.errstate(divide="ignore", invalid="ignore"):
            want = infer.normalize_raw_signal(r, "median").astype(np.float32)
        got = x[win_off[i]:win_off[i + 1]].reshape(-1)
        assert np.array_equal(got[:len(r)], want, equal_nan=True) and not got[len(r):].any(), (i, len(r))
    # post-processing over the same packing
    dev = torch.device("cuda", 0)
    s_off = win_off * 35
    total = int(s_off[-1])
    noise = rng.normal(size=total + 60)
    smooth = np.convolve(noise, np.ones(31) / 31.0, mode="same")[:total] * 6.0
    probs = (1.0 / (1.0 + np.exp(-smooth))).astype(np.float32)
    for i, n in enumerate(lens):
        probs[s_off[i] + n:s_off[i + 1]] = 1.0                                                          # padding must still come out as 0
    d_off, d_len = torch.from_numpy(s_off.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int64)).to(dev)
    starts, ends, labels = model.engine.postprocess_spans_device(torch.from_numpy(probs).to(dev), d_off, d_len, labels=True)
    labels = labels.cpu().numpy()
    want_starts, want_ends = [], []
    for i, n in enumerate(lens):
        p = probs[s_off[i]:s_off[i] + n]
        want = np.asarray(infer.correct_short(infer.class_from_threshold(p)))
        assert np.array_equal(labels[s_off[i]:s_off[i] + n], want) and not labels[s_off[i] + n:s_off[i + 1]].any(), i
        edges = np.flatnonzero(np.diff(np.concatenate(([0], want, [0]))))
        want_starts.extend((edges[0::2] + s_off[i]).tolist())
        want_ends.extend((edges[1::2] + s_off[i]).tolist())
    assert starts.tolist() == want_starts and ends.tolist() == want_ends and len(want_starts) > 1000
