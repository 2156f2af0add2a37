"""Config 5 (train step) on CPU: the torch restatement matches the oracle's forward, its gradients
match finite differences, and the TF-style optimizers follow their update formulas."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from catfish_amd.training import TorchResNetRNN, TFOptimizer, Trainer
from oracle import catfish_oracle as oracle


def _batch(n=8, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1.2, size=(n, 35)).astype(np.float32)
    y = np.repeat((np.arange(n) % 2)[:, None], 35, axis=1).astype(np.float32)   # uniform label per window
    return x, y


def test_torch_forward_matches_oracle(ckpt_weights):
    x, _ = _batch(6)
    net = TorchResNetRNN(ckpt_weights, 3, 2, dtype=torch.float64)
    with torch.no_grad():
        p = torch.sigmoid(net.logits(x)).numpy().reshape(-1)
    want = oracle.forward(x, ckpt_weights, np.float64)
    assert np.abs(p - want).max() < 1e-10


def test_loss_is_mean_sigmoid_cross_entropy(ckpt_weights):
    x, y = _batch(4)
    net = TorchResNetRNN(ckpt_weights, 3, 2, dtype=torch.float64)
    _, st = oracle.forward(x, ckpt_weights, np.float64, return_stages=True)
    z = st["logits"].reshape(4, 35)
    want = np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z))))
    assert abs(float(net.loss(x, y)) - want) < 1e-10


def test_gradients_match_finite_differences():
    w = oracle.random_weights(seed=5, n_layers=1, n_layers_res=1)
    x, y = _batch(3, seed=1)
    net = TorchResNetRNN(w, 1, 1, dtype=torch.float64)
    loss = net.loss(x, y)
    loss.backward()
    rng = np.random.default_rng(0)
    for name in ("conv1d_2/kernel", "batch_normalization_1/gamma",
                 "stack_bidirectional_rnn/cell_0/bidirectional_rnn/bw/gru_cell/candidate/kernel",
                 "final_fully_connected/bias"):
        p = net.params[name]
        idx = tuple(int(rng.integers(0, s)) for s in p.shape)
        eps = 1e-6
        with torch.no_grad():
            p[idx] += eps
            lp = float(net.loss(x, y))
            p[idx] -= 2 * eps
            lm = float(net.loss(x, y))
            p[idx] += eps
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - float(p.grad[idx])) < 1e-6 * max(1.0, abs(fd)), name
    assert net.params["batch_normalization/moving_mean"].grad is None      # BN statistics do not train


@pytest.mark.parametrize("choice", ["Adam", "RMSProp"])
def test_tf_optimizer_formulas(choice):
    p = {"w": torch.tensor([1.0, -2.0], dtype=torch.float64, requires_grad=True)}
    opt = TFOptimizer(p, choice, 0.1)
    w = np.array([1.0, -2.0]); m = np.zeros(2); v = np.zeros(2); ms = np.ones(2)
    for t in range(1, 4):
        g = np.array([0.5 * t, -1.0])
        p["w"].grad = torch.tensor(g)
        opt.step()
        if choice == "Adam":
            m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
            w = w - 0.1 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        else:
            ms = 0.9 * ms + 0.1 * g * g
            w = w - 0.1 * g / np.sqrt(ms + 1e-10)
        assert np.allclose(p["w"].detach().numpy(), w, atol=1e-12)
    with pytest.raises(ValueError):
        TFOptimizer(p, "SGD", 0.1)


def test_training_reduces_loss_and_keeps_shapes():
    w = oracle.random_weights(seed=2)
    x, y = _batch(16, seed=3)
    tr = Trainer(w, 3, 2, "Adam", 1e-3, keep_prob=1.0, device="cpu", seed=0)
    losses = [tr.train_step(x, y) for _ in range(12)]
    assert losses[-1] < losses[0] - 1e-3, losses
    nw = tr.net.numpy_weights()
    assert sorted(nw) == sorted(w) and all(nw[k].shape == np.asarray(w[k]).shape for k in w)
    assert np.array_equal(nw["batch_normalization/moving_variance"], w["batch_normalization/moving_variance"])


def test_dropout_only_on_outputs_is_seeded():
    w = oracle.random_weights(seed=2, n_layers=1, n_layers_res=1)
    x, y = _batch(4)
    a = Trainer(w, 1, 1, "RMSProp", 1e-3, keep_prob=0.8, device="cpu", seed=7).train_step(x, y)
    b = Trainer(w, 1, 1, "RMSProp", 1e-3, keep_prob=0.8, device="cpu", seed=7).train_step(x, y)
    c = Trainer(w, 1, 1, "RMSProp", 1e-3, keep_prob=1.0, device="cpu", seed=7).train_step(x, y)
    assert a == b and a != c


def test_trainer_has_no_silent_cpu_fallback():
    """Without a HIP device the default Trainer refuses to run; the torch-CPU mode is opt-in (device='cpu')."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    w = oracle.random_weights(seed=1, n_layers=1, n_layers_res=1)
    with pytest.raises(RuntimeError, match="no HIP device"):
        Trainer(w, 1, 1, "Adam", 1e-3, keep_prob=1.0)


def test_adam_step_count_survives_a_long_run():
    """tf.train.Saver stores beta1_power = 0.9^(t+1) and beta2_power = 0.999^(t+1) as float32.  After ~830 steps the first
    is denormal / zero, so the step count is restored from the second (and a huge count stands in when that is gone
    too: the bias corrections are 1 by then)."""
    p = {"w": torch.tensor([1.0, -2.0], dtype=torch.float32, requires_grad=True)}
    for t in (0, 5, 500, 3000, 50000):
        opt = TFOptimizer(p, "Adam", 0.1)
        opt.t.fill_(t)
        state = opt.state_tf()
        assert state["optimizer/beta1_power"].dtype == np.float32
        fresh = TFOptimizer(p, "Adam", 0.1)
        fresh.load_state_tf(state)
        assert abs(float(fresh.t) - t) <= max(1, t * 1e-3), (t, float(fresh.t))
    gone = TFOptimizer(p, "Adam", 0.1)
    gone.load_state_tf({"optimizer/beta1_power": np.float32(0.0), "optimizer/beta2_power": np.float32(0.0)})
    assert float(gone.t) >= 1e5


def test_anysize_pack_maps_follow_the_fragment_formula():
    """catfish_amd/anysize_train.pack_maps (vectorised numpy) against the definition of the A-fragment layout written as plain
    loops (include/catfish_hip.h, cf_gru_anysize_train_forward): component i of lane l of block (mo, kb) holds
    W[in = 16 kb + 4 (l >> 4) + i][out = 16 mo + (l & 15)], x rows zero-padded to whole blocks, r / u scaled by -log2 e and the
    candidate by 2 log2 e; the transposed packs of the backward chain likewise."""
    torch = pytest.importorskip("torch")
    from catfish_amd import anysize_train as at
    h, cin = 32, 20                                   # cin not a multiple of 16: exercises the zero padding
    rng = np.random.default_rng(0)
    wg = rng.normal(size=(cin + h, 2 * h)).astype(np.float32)
    wc = rng.normal(size=(cin + h, h)).astype(np.float32)
    bg = rng.normal(size=2 * h).astype(np.float32)
    bc = rng.normal(size=h).astype(np.float32)
    src = torch.from_numpy(np.concatenate([wg.reshape(-1), wc.reshape(-1), bg, bc, np.zeros(1, np.float32)]))
    w_idx, w_scale, b_idx, b_scale, wt_idx = at.pack_maps(h, cin, "cpu")
    h16, kbx = h // 16, (cin + 15) // 16
    kb_all = kbx + h16
    got_w = (src[w_idx] * w_scale).numpy().reshape(3, h16, kb_all, 64, 4)
    got_b = (src[b_idx] * b_scale).numpy().reshape(3, h16, 64, 4)
    got_t = src[wt_idx].numpy()
    gate_scale, cand_scale = -1.4426950408889634, 2 * 1.4426950408889634
    for gate in range(3):
        mat = wg[:, :h] if gate == 0 else (wg[:, h:] if gate == 1 else wc)
        bias = bg[:h] if gate == 0 else (bg[h:] if gate == 1 else bc)
        sc = gate_scale if gate < 2 else cand_scale
        for mo in range(h16):
            for lane in range(64):
                for j in range(4):
                    assert got_b[gate, mo, lane, j] == np.float32(bias[16 * mo + 4 * (lane >> 4) + j] * np.float32(sc))
                for kb in range(kb_all):
                    for i in range(4):
                        inn, out = 16 * kb + 4 * (lane >> 4) + i, 16 * mo + (lane & 15)
                        if inn < 16 * kbx:
                            want = mat[inn, out] if inn < cin else 0.0
                        else:
                            want = mat[cin + inn - 16 * kbx, out]
                        assert got_w[gate, mo, kb, lane, i] == np.float32(np.float32(want) * np.float32(sc)), (gate, mo, kb, lane, i)
    wct = got_t[:h16 * h16 * 256].reshape(h16, h16, 64, 4)
    wgt = got_t[h16 * h16 * 256:].reshape(h16, 2 * h16, 64, 4)
    for mo in range(h16):
        for lane in range(0, 64, 7):
            for i in range(4):
                for kb in range(h16):
                    assert wct[mo, kb, lane, i] == wc[cin + 16 * mo + (lane & 15), 16 * kb + 4 * (lane >> 4) + i]
                for kb in range(2 * h16):
                    assert wgt[mo, kb, lane, i] == wg[cin + 16 * mo + (lane & 15), 16 * kb + 4 * (lane >> 4) + i]
