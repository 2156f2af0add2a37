"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; ``catfish_amd`` itself must not (it fails
loudly when the HIP library is missing instead of falling back to this code).

What this is
------------
A numpy restatement (fp64 = ground truth, fp32 = stand-in for the reference's
TensorFlow-1.10 CPU execution) of the homopolymer-calling forward pass of
MMAThijssen/catfish and of the pure-Python pre/post-processing around it.
Each function cites the reference file:line it follows (paths relative to the
reference repository root).

Pinning status
--------------
* Network forward (``forward``): **graph pinned, op arithmetic unpinned**.  The
  reference ships the MetaGraphDef its TensorFlow-1.10 session executed
  (ckpnt-30000.meta).  ``oracle/tf_graph.py`` decodes it (wire-format parser,
  nothing executed) and interprets its inference subgraph node by node;
  ``forward`` agrees with that interpretation to 1e-10 in fp64 on the bundled
  checkpoint and on random variables, stage by stage
  (tests/test_reference_graph.py, vectors in tests/golden/graph_golden.npz from
  tests/golden/make_graph_golden.py).  That pins everything the reference's own
  file can pin: op order, every attribute (SAME padding, unit strides, NHWC,
  split into [r, u], reset applied to the state BEFORE the candidate matmul,
  ``h' = u*h + (1-u)*c``, bw = reverse -> run -> reverse, concat [fw, bw],
  unfused inference batch norm with float32 epsilon 1e-3, dropout wiring), and
  which variable feeds which op.  What stays a restatement is the arithmetic of
  the individual TF op kernels (Conv2D, MatMul, Sigmoid, Tanh ...): TensorFlow
  1.10.0 is third-party, not importable here and not installable, and the
  reference ships no tests or recorded outputs of a real TF run -- in that
  sense parity with an actual TF execution remains **parity unpinned**.
* Checkpoint reader: pinned by the per-tensor masked CRC-32C values stored in
  ckpnt-30000.index (tests/test_checkpoint.py).
* Pre/post-processing (``normalize_raw_signal``, ``pad_and_window``,
  ``class_from_threshold``, ``correct_short``, ``hp_in_pred``, ``center_hp``):
  pinned by golden vectors produced by executing the reference's own functions
  (tests/golden/make_postproc_golden.py; fixtures in tests/golden/).
"""
from __future__ import annotations

import numpy as np

WINDOW = 35          # rnn_class.py:27
BN_EPS = 1e-3        # tf.layers.batch_normalization default; ckpt:meta batchnorm/add/y


# --------------------------------------------------------------------------- weights helpers
def conv_names(j):
    return "conv1d" if j == 0 else "conv1d_%d" % j


def bn_names(j):
    return "batch_normalization" if j == 0 else "batch_normalization_%d" % j


def gru_prefix(layer, direction):
    return "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell" % (layer, direction)


def random_weights(seed=0, layer_size=64, n_layers=3, layer_size_res=32, n_layers_res=2,
                   dtype=np.float32):
    """Random weights with the checkpoint's tensor names/shapes.

    Initialisers follow TF defaults but BN statistics and biases
    are perturbed so that every term of the graph is exercised by tests.
    """
    rng = np.random.default_rng(seed)
    w = {}

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    j = 0
    cin = 1
    for _ in range(n_layers_res):
        for k in (1, 1, 3, 1):
            c_in = cin if j % 4 in (0, 1) else layer_size_res
            w[conv_names(j) + "/kernel"] = glorot((k, c_in, layer_size_res), k * c_in, k * layer_size_res)
            w[conv_names(j) + "/bias"] = rng.normal(0, 0.1, layer_size_res).astype(dtype)
            w[bn_names(j) + "/gamma"] = rng.uniform(0.5, 1.5, layer_size_res).astype(dtype)
            w[bn_names(j) + "/beta"] = rng.normal(0, 0.1, layer_size_res).astype(dtype)
            w[bn_names(j) + "/moving_mean"] = rng.normal(0, 0.1, layer_size_res).astype(dtype)
            w[bn_names(j) + "/moving_variance"] = rng.uniform(0.5, 1.5, layer_size_res).astype(dtype)
            j += 1
        cin = layer_size_res
    c_in = layer_size_res if n_layers_res > 0 else 1
    for layer in range(n_layers):
        for d in ("fw", "bw"):
            p = gru_prefix(layer, d)
            k = c_in + layer_size
            w[p + "/gates/kernel"] = glorot((k, 2 * layer_size), k, 2 * layer_size)
            w[p + "/gates/bias"] = (1.0 + rng.normal(0, 0.1, 2 * layer_size)).astype(dtype)
            w[p + "/candidate/kernel"] = glorot((k, layer_size), k, layer_size)
            w[p + "/candidate/bias"] = rng.normal(0, 0.1, layer_size).astype(dtype)
        c_in = 2 * layer_size
    w["final_fully_connected/kernel"] = glorot((2 * layer_size, 1), 2 * layer_size, 1)
    w["final_fully_connected/bias"] = rng.normal(0, 0.1, 1).astype(dtype)
    return w


# --------------------------------------------------------------------------- network forward
def _sigmoid(x):
    # tf.nn.sigmoid; written to avoid overflow warnings in fp32
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(x, filters, K, padding="same") -- resnet_class.py:60,64,69,74.

    x [N, T, Cin]; kernel [K, Cin, Cout]; cross-correlation with zero padding
    (K-1)//2 left, K-1-(K-1)//2 right, applied inside each window.
    """
    n, t, _ = x.shape
    k = kernel.shape[0]
    left = (k - 1) // 2
    xp = np.zeros((n, t + k - 1, x.shape[2]), dtype=x.dtype)
    xp[:, left:left + t, :] = x
    out = np.zeros((n, t, kernel.shape[2]), dtype=x.dtype)
    for i in range(k):
        out += xp[:, i:i + t, :] @ kernel[i]
    return out + bias


def batch_norm_inference(x, gamma, beta, mean, var):
    """tf.layers.batch_normalization(x) with training never passed -- resnet_class.py:61,65,70,75.

    Unfused form of ckpt:meta: inv = rsqrt(var + eps) * gamma; x*inv + (beta - mean*inv).
    """
    dt = x.dtype
    inv = (gamma / np.sqrt(var + dt.type(BN_EPS))).astype(dt)
    return x * inv + (beta - mean * inv).astype(dt)


def residual_block(x, w, j0):
    """resnet_class.py:44-82.  j0 = index of the block's first conv/bn pair."""
    def cb(inp, j):
        y = conv1d_same(inp, w[conv_names(j) + "/kernel"], w[conv_names(j) + "/bias"])
        return batch_norm_inference(y, w[bn_names(j) + "/gamma"], w[bn_names(j) + "/beta"],
                                    w[bn_names(j) + "/moving_mean"], w[bn_names(j) + "/moving_variance"])
    sc = cb(x, j0)                                  # :60-61 shortcut, no relu
    o1 = np.maximum(cb(x, j0 + 1), 0)               # :64-66
    o2 = np.maximum(cb(o1, j0 + 2), 0)              # :69-71
    o3 = np.maximum(cb(o2, j0 + 3), 0)              # :74-76
    return np.maximum(o3 + sc, 0)                   # :79-80


def gru_direction(x, wg, bg, wc, bc, reverse):
    """One direction of one layer: tf.contrib.rnn.GRUCell(64) unrolled over T with zero state.

    rnn_class.py:146,167-171.  x [N, T, Cin] -> [N, T, H] (already re-reversed for bw).
    """
    n, t, cin = x.shape
    h_size = wc.shape[1]
    h = np.zeros((n, h_size), dtype=x.dtype)
    out = np.empty((n, t, h_size), dtype=x.dtype)
    steps = range(t - 1, -1, -1) if reverse else range(t)
    for s in steps:
        xt = x[:, s, :]
        gates = _sigmoid(np.concatenate([xt, h], axis=1) @ wg + bg)
        r, u = gates[:, :h_size], gates[:, h_size:]
        c = np.tanh(np.concatenate([xt, r * h], axis=1) @ wc + bc)
        h = u * h + (1 - u) * c
        out[:, s, :] = h
    return out


def forward(x, w, dtype=np.float64, n_layers=3, n_layers_res=2, return_stages=False):
    """ResNetRNN forward: [N, 35, 1] (or [N, 35]) -> probabilities [N*35] (float64 like RNN.infer).

    resnet_class.py:17-25 -> rnn_class.py:165-183 -> rnn_class.py:84 -> rnn_class.py:213-219.
    """
    dtype = np.dtype(dtype)
    x = np.asarray(x)
    if x.ndim == 2:
        x = x[:, :, None]
    a = x.astype(dtype)
    wd = {k: np.asarray(v).astype(dtype) for k, v in w.items()}
    stages = {}
    for d in range(n_layers_res):
        a = residual_block(a, wd, 4 * d)
        stages["res%d" % d] = a
    for layer in range(n_layers):
        outs = []
        for dname, rev in (("fw", False), ("bw", True)):
            p = gru_prefix(layer, dname)
            outs.append(gru_direction(a, wd[p + "/gates/kernel"], wd[p + "/gates/bias"],
                                      wd[p + "/candidate/kernel"], wd[p + "/candidate/bias"], rev))
        a = np.concatenate(outs, axis=2)
        stages["gru%d" % layer] = a
    logits = a.reshape(-1, a.shape[2]) @ wd["final_fully_connected/kernel"] + wd["final_fully_connected/bias"]
    probs = _sigmoid(logits).reshape(-1).astype(np.float64)
    if return_stages:
        stages["logits"] = logits.reshape(-1)
        return probs, stages
    return probs


# --------------------------------------------------------------------------- pre-processing
def normalize_raw_signal(raw):
    """infer.py:96-105 (median / raw MAD, no 1.4826 factor)."""
    raw = np.asarray(raw)
    shift = np.median(raw)
    scale = np.median(np.abs(raw - shift))
    return (raw - shift) / scale


def padding_size(length, window=WINDOW):
    """infer.py:32-36 -- a full extra window when the length is already a multiple."""
    if length % window != 0:
        return window - (length - (length // window) * window)
    return 35


def pad_and_window(raw, window=WINDOW):
    """infer.py:31-43,108-124: zero-pad then reshape to [N, window, 1]."""
    pad = padding_size(len(raw), window)
    padded = np.hstack((np.asarray(raw), np.array(pad * [0])))
    return np.reshape(padded, (-1, window, 1)), pad


# --------------------------------------------------------------------------- post-processing
def class_from_threshold(scores, threshold=0.5):
    """infer.py:128-138."""
    return [1 if y >= threshold else 0 for y in scores]


def _rle(values):
    runs = []  # [value, length, start]
    for p, v in enumerate(values):
        if runs and runs[-1][0] == v:
            runs[-1][1] += 1
        else:
            runs.append([v, 1, p])
    return runs


def correct_short(predictions, threshold=15):
    """infer.py:174-198: positive runs shorter than threshold become 0."""
    runs = _rle(list(predictions))
    for r in runs:
        if r[0] != 0 and r[1] < threshold:
            r[0] = 0
    if not runs:
        return np.zeros(0, dtype=np.int64)
    return np.concatenate([np.repeat(r[0], r[1]) for r in runs])


def hp_in_pred(predictions, extension_left=11, extension_right=16, label=1):
    """infer.py:141-162: every run of `label` -> [start - 11, start + len + 16]."""
    runs = _rle(list(predictions))
    return [[r[2] - extension_left, r[2] + r[1] + extension_right] for r in runs if r[0] == label]


def center_hp(merged_positions, len_read, chunk_size=1000):
    """catfish/catfish:121-135 (mutates the last entry in place, quirks included)."""
    len_hp = merged_positions[-1][-1] - merged_positions[-1][0]
    if len_hp < chunk_size:
        left = (chunk_size - len_hp) // 2
        right = (chunk_size - len_hp) - left
        merged_positions[-1][0] -= left
        merged_positions[-1][1] += right
        if merged_positions[-1][0] < 0:
            merged_positions[-1][1] -= merged_positions[-1][0]
            merged_positions[-1][0] = 0
        if merged_positions[-1][1] > len_read:
            merged_positions[-1][0] -= len_read - merged_positions[-1][1]
            merged_positions[-1][1] = len_read
    return merged_positions


def infer_read(signal_norm, w, dtype=np.float32):
    """infer.py:31-51 on an already-normalised signal: -> (hp spans, len(labels), scores)."""
    raw_in, pad = pad_and_window(signal_norm)
    scores = forward(raw_in, w, dtype=dtype)
    scores = scores[:-pad]
    labels = correct_short(class_from_threshold(scores))
    return hp_in_pred(labels), len(labels), scores


# --------------------------------------------------------------------------- synthetic data
def synthetic_dac(n_reads, length, seed=0):
    """Seeded int16 DAC squiggles: piecewise-constant levels N(500,60^2), dwell Geometric(1/9),
    noise N(0,8^2), clipped to [0,2047]."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_reads, length), dtype=np.int16)
    for i in range(n_reads):
        n_ev = length // 4 + 8
        dwell = rng.geometric(1.0 / 9.0, size=n_ev)
        while dwell.sum() < length:
            dwell = np.concatenate([dwell, rng.geometric(1.0 / 9.0, size=n_ev)])
        levels = rng.normal(500.0, 60.0, size=len(dwell))
        sig = np.repeat(levels, dwell)[:length] + rng.normal(0.0, 8.0, size=length)
        out[i] = np.clip(np.rint(sig), 0, 2047).astype(np.int16)
    return out
