"""Host logic of the training / validation driver (reference networks/train_validate.py) and of the model report
(rnn_class.py:264-270, resnet_class.py:28-32) -- no GPU needed."""
import os

import numpy as np
import pytest

from catfish_amd import metrics, neural_network, train_validate as tv
from catfish_amd.resnet_class import ResNetRNN
from catfish_amd.rnn_class import RNN, sigmoid_cross_entropy_from_logits
from conftest import REFERENCE, has_reference


def test_padding_of_the_training_driver_differs_from_infer():
    x, pad = tv.padding(np.arange(70.0))
    assert x.shape == (2, 35, 1) and pad == 0                 # exact multiple: NO extra window (train_validate.py:50-63)
    x, pad = tv.padding(np.arange(71.0))
    assert x.shape == (3, 35, 1) and pad == 34 and np.all(x.reshape(-1)[71:] == 0)


def _brute_windows(raw, labels, width, lessen):
    """TrainingRead.get_pos / get_neg (TrainingRead.py:226-257) loop for loop (all negatives instead of a sample)."""
    labels = list(labels)
    width_l = width // 2
    width_r = width - width_l
    final = len(labels)
    hits = [i for i in range(width_l, final - width_r) if labels[i] == 1]
    pos = []
    for ch in range(0, len(hits), lessen):
        s, e = hits[ch] - width_l, hits[ch] + width_r + 1
        if len(labels[s:e]) == labels[s:e].count(1):
            pos.append(raw[s:e])
    neg = []
    for i in [i for i in range(width_l, final - width_r) if labels[i] == 0]:
        s, e = i - width_l, i + width_r + 1
        if len(labels[s:e]) == labels[s:e].count(0):
            neg.append(raw[s:e])
    return pos, neg


@pytest.mark.parametrize("width,lessen", [(34, 1), (34, 3), (35, 2), (4, 1)])
def test_uniform_label_windows_match_the_reference_sampler(width, lessen):
    rng = np.random.default_rng(width + lessen)
    labels = np.repeat(rng.integers(0, 2, size=60), rng.integers(1, 60, size=60))
    raw = rng.normal(size=len(labels))
    pos, neg = tv.windows_from_labelled_read(raw, labels, width, lessen)
    bpos, bneg = _brute_windows(raw, labels, width, lessen)
    assert len(pos) == len(bpos) and all(np.array_equal(a, b) for a, b in zip(pos, bpos))
    assert len(neg) == len(bneg) and all(np.array_equal(a, b) for a, b in zip(neg, bneg))
    assert all(len(w) == width + 1 for w in pos + neg)


def test_balanced_batches_have_the_reference_shape():
    """ExampleDb.get_training_set (ExampleDb.py:50-83): size // 2 all-positive + the rest all-negative windows."""
    db = tv.synthetic_example_db(n_reads=2, read_len=12000, seed=1)
    assert db.nb_pos >= 128 and db.nb_neg >= 128
    x, y, pos = db.get_training_set(256, ratio=2)
    assert len(x) == len(y) == 256 and pos == 128 * 35
    assert all(len(w) == 35 for w in x) and all(len(set(l)) == 1 for l in y)
    assert sorted(l[0] for l in y) == [0] * 128 + [1] * 128
    assert [l[0] for l in y] != sorted(l[0] for l in y)                        # shuffled
    sx = tv.reshape_input(x, 35, 1)
    sy = tv.reshape_input(y, 35, 1)
    assert sx.shape == sy.shape == (256, 35, 1)
    x2, _, pos2 = db.get_training_set(255, ratio=2)
    assert pos2 == 127 * 35 and len(x2) == 255


def test_random_hyperparameters_follow_the_reference_draws():
    np.random.seed(3)
    got = tv.generate_random_hyperparameters("ResNetRNN")
    np.random.seed(3)                                                          # networks/train_validate.py:89-109
    lr = 10 ** np.random.randint(-4, 0)
    opt = np.random.choice(["Adam", "RMSProp"])
    ls = np.random.choice([16, 32, 64, 128, 256])
    nl = np.random.randint(1, 6)
    bs = np.random.choice([128, 256, 512])
    kp = round(np.random.uniform(0.2, 0.8), 1)
    np.random.randint(1, 12)
    lsr = np.random.choice([16, 32, 64, 128, 256])
    assert got == {"batch_size": bs, "optimizer_choice": opt, "learning_rate": lr, "layer_size": ls, "n_layers": nl,
                   "keep_prob": kp, "layer_size_res": lsr, "n_layers_res": nl}   # n_layers_res = n_layers: kept quirk
    assert set(tv.generate_random_hyperparameters("RNN")) == {"batch_size", "optimizer_choice", "learning_rate",
                                                               "layer_size", "n_layers", "keep_prob"}


def test_scores_from_confusion_counts():
    assert metrics.precision_recall(3, 1, 2) == (0.75, 0.6)
    assert metrics.precision_recall(0, 0, 0) == (0, 0)
    assert metrics.calculate_accuracy(1, 1, 1, 1) == 0.5 and metrics.calculate_accuracy(0, 0, 0, 0) == 0
    assert metrics.f1(0.5, 0.5) == 0.5 and metrics.f1(0, 0) == 0
    assert metrics.weighted_f1(0.5, 0.5, 10, 40) == 0.125 and metrics.weighted_f1(0, 0, 1, 2) == 0
    assert metrics.class_from_threshold([0.1, 0.5, 0.9], 0.5) == [0, 1, 1]


def test_loss_from_logits_survives_saturation():
    """tf.losses.sigmoid_cross_entropy on logits (rnn_class.py:74-79): a confident wrong sample costs |z|, not
    -log(tiny) as a loss rebuilt from saturated fp32 probabilities would."""
    z = np.array([40.0, -40.0, 17.0, 0.0, -3.0, 100.0])
    y = np.array([0.0, 1.0, 1.0, 1.0, 0.0, 1.0])
    want = np.mean([40.0, 40.0, np.log1p(np.exp(-17.0)), np.log(2.0), np.log1p(np.exp(-3.0)), 0.0])
    assert abs(sigmoid_cross_entropy_from_logits(z, y) - want) < 1e-12
    p32 = (1.0 / (1.0 + np.exp(-z.astype(np.float32)))).astype(np.float32)
    assert p32[0] == 1.0                                                       # the probability really saturates in fp32


def test_save_info_writes_the_reference_report(tmp_path, monkeypatch, hp):
    """save=True (rnn_class.py:43-46,100-118): first free <cwd>/<model type>_<n>; report text as the reference
    writes it (rnn_class.py:264-270 + resnet_class.py:28-32) and as retrieve_hyperparams reads it back."""
    monkeypatch.chdir(tmp_path)
    m = ResNetRNN(save=True, **hp)
    assert m.model_path == str(tmp_path / "ResNet-RNN_0") and os.path.isdir(m.model_path)
    text = open(m.model_path + ".txt").read()
    assert text == ("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\nlearning_rate: 0.001\n"
                    "layer_size: 64\nn_layers: 3\nkeep_prob: 0.8\nlayer_size_res: 32\nn_layers_res: 2\n\n")
    assert open(os.path.join(m.model_path, "ResNetRNN.txt")).read() == text
    if has_reference():                                                        # the shipped model's own report header
        shipped = open(os.path.join(REFERENCE, "catfish", "ResNetRNN", "ResNetRNN.txt")).read()
        assert shipped.startswith(text)
    assert neural_network.retrieve_hyperparams(os.path.join(m.model_path, "ResNetRNN.txt")) == hp
    m2 = ResNetRNN(save=True, **hp)
    assert m2.model_path.endswith("ResNet-RNN_1")
    r = RNN(save=True, batch_size=128, optimizer_choice="Adam", learning_rate=0.01, layer_size=64, n_layers=2, keep_prob=0.5)
    assert r.model_path.endswith("biGRU-RNN_0")
    assert open(r.model_path + ".txt").read() == ("MODEL TYPE: biGRU-RNN\n\nbatch_size: 128\noptimizer_choice: Adam\n"
                                                  "learning_rate: 0.01\nlayer_size: 64\nn_layers: 2\nkeep_prob: 0.5\n")
    with pytest.raises(RuntimeError):
        ResNetRNN(**hp).save_info()                                           # no model directory claimed
