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


# ---------------------------------------------------------------- validate(): the reference's own report, byte for byte
class _StubNetwork(object):
    """The closed-form 'network' of tests/golden/validate_stub.py behind the surface ``validate`` uses."""
    window, n_inputs, n_outputs, model_type = 35, 1, 1, "ResNet-RNN"

    def __init__(self, a, b, counters):
        self.a, self.b = a, b
        self.tp, self.fp, self.tn, self.fn = counters
        self.calls = []

    def score_windows(self, windows):
        from golden import validate_stub as stub
        self.calls.append(np.asarray(windows).shape)
        logits = stub.stub_logits(np.asarray(windows).reshape(-1), self.a, self.b)
        return stub.stub_probs(logits), logits


def _validate_golden():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "validate_golden.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture()
def golden_npz_reads(tmp_path):
    paths = []
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "validate_golden_reads.npz")) as z:
        for i in range(len(z.files) // 2):
            paths.append(str(tmp_path / ("read_%02d.npz" % i)))
            np.savez(paths[-1], raw=z["raw_%02d" % i], base_labels=z["labels_%02d" % i])
    return paths


@pytest.mark.parametrize("case", _validate_golden(), ids=lambda c: c["name"])
def test_validate_reproduces_the_reference_report(case, golden_npz_reads, tmp_path, monkeypatch, capsys):
    """tests/golden/make_validate_golden.py ran the reference's validate / padding / test_network / metrics functions
    (lifted from its files) on these reads; the packed implementation must write the same bytes, print the same lines,
    return the same numbers and leave the same counters -- incl. the max_number read that is counted but not averaged,
    the p = 0.5 samples and zero tails that are called positive, and counters carried in from before."""
    import random
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    monkeypatch.chdir(tmp_path)
    net = _StubNetwork(case["a"], case["b"], case["counters_before"])
    if case["random_seed"] is not None:
        random.seed(case["random_seed"])
    capsys.readouterr()
    got = tv.validate(net, golden_npz_reads, case["max_seq_length"], "some/dir/" + case["name"],
                      case["validation_start"], case["max_number"])
    assert capsys.readouterr().out == case["printed"]
    assert open(case["name"] + ".txt").read() == case["report"]
    assert [float(v) for v in got] == case["returned"]
    assert [net.tp, net.fp, net.tn, net.fn] == case["counters_after"] == [0, 0, 0, 0]
    assert len(net.calls) == 1                                                 # ONE packed launch per validation round


def test_per_read_surface_agrees_with_the_packed_round(golden_npz_reads):
    """RNN.test_network (one read per call, rnn_class.py:222-261) and the packed round count the same things."""
    from catfish_amd.rnn_class import RNN

    class PerRead(_StubNetwork):
        test_network = RNN.test_network

    a, b = 0.75, 0.375
    net = PerRead(a, b, (0, 0, 0, 0))
    signals, labels = tv.select_validation_stretches(golden_npz_reads, 35, 0, "complete", 856)
    accs, losses = [], []
    for s, l in zip(signals, labels):
        x, pad = tv.padding(s)
        y, _ = tv.padding(l)
        acc, loss = net.test_network(x, y, "r", "f", pad)
        accs.append(acc)
        losses.append(loss)
    x, y, bounds, tails = tv.pack_validation_windows(signals, labels, 35)
    pk = _StubNetwork(a, b, (0, 0, 0, 0))
    acc, loss, counts = tv.score_validation_batch(*pk.score_windows(x), y, bounds, tails)
    assert counts == (net.tp, net.fp, net.tn, net.fn)
    assert np.array_equal(acc, np.float32(accs)) and np.array_equal(loss, np.float32(losses))


def test_validate_refuses_what_the_reference_cannot_run(golden_npz_reads, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    net = _StubNetwork(1.0, 0.0, (0, 0, 0, 0))
    with pytest.raises(ZeroDivisionError):                                     # every read too short (:268 divides by 0)
        tv.validate(net, golden_npz_reads, 35000, "m", 0, 856)
    with pytest.raises(ValueError):
        tv.validate(net, golden_npz_reads, 350, "m", "middle", 856)
    assert not os.path.exists("m.txt") and net.calls == []
