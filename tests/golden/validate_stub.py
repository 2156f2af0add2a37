"""A deterministic stand-in for the network, shared by make_validate_golden.py (where it plays ``tf.Session`` under the
reference's own ``validate`` / ``test_network``) and by tests/test_train_validate.py (where it plays the HIP engine
under ``catfish_amd.train_validate.validate``).  Own code, no reference text.

logit(x) = float32(x) * a + b in float32 (two IEEE operations: the same bits on every machine); the reads of the
golden hold multiples of 0.25, so ``a = 0.75, b = -0.375`` puts samples EXACTLY on p = 0.5, where the reference's two
classification rules disagree (``c >= threshold`` says 1, ``tf.round`` says 0: rnn_class.py:235 vs :85).
"""
import numpy as np


def stub_logits(x, a, b):
    x32 = np.asarray(x, dtype=np.float32)
    return (x32 * np.float32(a) + np.float32(b)).astype(np.float32)


def stub_probs(logits32):
    """tf.nn.sigmoid in float32 (evaluated in double, rounded once)."""
    return (1.0 / (1.0 + np.exp(-logits32.astype(np.float64)))).astype(np.float32)


def stub_accuracy_loss(probs32, logits32, labels):
    """What ``sess.run([accuracy, loss])`` returns (rnn_class.py:74-88): float32 scalars.
    accuracy = mean(round_half_even(p) == y) -- an exact count divided once in float32;
    loss = mean(max(z,0) - z*y + log1p(exp(-|z|))) -- summed in double, rounded once to float32."""
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    p = probs32.astype(np.float64).reshape(-1)
    z = logits32.astype(np.float64).reshape(-1)
    acc = np.float32(np.count_nonzero(np.round(p) == y)) / np.float32(len(y))
    loss = np.float32(np.sum(np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z)))) / len(y))
    return acc, loss
