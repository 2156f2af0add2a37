"""The oracle against the committed fixtures, and its internal consistency (fp32 vs fp64)."""
import json
import os

import numpy as np

from oracle import catfish_oracle as oracle
from conftest import GOLDEN


def test_oracle_reproduces_golden_read(ckpt_weights, golden_read):
    p64 = oracle.forward(golden_read["x"], ckpt_weights, np.float64)
    assert np.allclose(p64, golden_read["probs_fp64"], atol=1e-12)
    p32 = oracle.forward(golden_read["x"], ckpt_weights, np.float32)
    assert np.abs(p32 - p64).max() < 5e-6
    assert p64.shape == (118 * 35,)
    assert int(golden_read["pad"]) == 34


def test_oracle_window_independence(ckpt_weights, golden_read):
    """Windows are independent: a subset of windows gives the same per-window output."""
    x = golden_read["x"]
    full = oracle.forward(x, ckpt_weights, np.float64).reshape(-1, 35)
    part = oracle.forward(x[40:45], ckpt_weights, np.float64).reshape(-1, 35)
    assert np.allclose(full[40:45], part, atol=1e-12)


def test_oracle_gru_semantics_against_scalar_loop():
    """TF-1 GRUCell: r,u = split(sigmoid([x,h] Wg + bg)); c = tanh([x, r*h] Wc + bc); h' = u h + (1-u) c."""
    rng = np.random.default_rng(0)
    cin, h = 3, 4
    wg = rng.normal(size=(cin + h, 2 * h)); bg = rng.normal(size=2 * h)
    wc = rng.normal(size=(cin + h, h)); bc = rng.normal(size=h)
    x = rng.normal(size=(1, 5, cin))
    got = oracle.gru_direction(x, wg, bg, wc, bc, reverse=False)[0]
    state = np.zeros(h)
    for t in range(5):
        g = 1 / (1 + np.exp(-(np.concatenate([x[0, t], state]) @ wg + bg)))
        r, u = g[:h], g[h:]
        c = np.tanh(np.concatenate([x[0, t], r * state]) @ wc + bc)
        state = u * state + (1 - u) * c
        assert np.allclose(got[t], state)
    rev = oracle.gru_direction(x, wg, bg, wc, bc, reverse=True)[0]
    fwd_on_flipped = oracle.gru_direction(x[:, ::-1], wg, bg, wc, bc, reverse=False)[0][::-1]
    assert np.allclose(rev, fwd_on_flipped)


def test_oracle_conv_same_padding_is_window_local():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 35, 3))
    k = rng.normal(size=(3, 3, 5)); b = rng.normal(size=5)
    y = oracle.conv1d_same(x, k, b)
    assert np.allclose(y[0, 0], x[0, 0] @ k[1] + x[0, 1] @ k[2] + b)          # left edge: zero pad
    assert np.allclose(y[0, 34], x[0, 33] @ k[0] + x[0, 34] @ k[1] + b)       # right edge
    assert np.allclose(y[1, 10], x[1, 9] @ k[0] + x[1, 10] @ k[1] + x[1, 11] @ k[2] + b)


def test_oracle_postprocessing_against_reference_goldens():
    with open(os.path.join(GOLDEN, "postproc_golden.json")) as fh:
        g = json.load(fh)
    for c in g["postproc"]:
        labels = oracle.class_from_threshold(c["scores"])
        assert labels == c["labels"]
        corrected = oracle.correct_short(labels)
        assert corrected.tolist() == c["corrected"]
        assert oracle.hp_in_pred(corrected) == c["spans"]
    for c in g["normalize"]:
        out = oracle.normalize_raw_signal(np.array(c["raw"], dtype=np.int16))
        assert np.array_equal(out, np.array(c["out"]))
    for c in g["padding"]:
        assert oracle.padding_size(c["length"]) == c["padding_size"]
        x, pad = oracle.pad_and_window(np.arange(c["length"], dtype=np.float64))
        assert list(x.shape) == c["shape"] and pad == c["padding_size"]
    for c in g["center_hp"]:
        out = oracle.center_hp([[0, 5], list(c["in"])], c["len_read"], c["chunk_size"])
        assert out[-1] == c["out"]
