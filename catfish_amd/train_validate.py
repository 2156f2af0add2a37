"""Training / validation driver -- the reference's networks/train_validate.py on the MI355X model objects.

Same function names, arguments and report text as the reference (``train_and_validate``
networks/train_validate.py:114-185, ``validate`` :188-295, ``padding`` :50-63, ``reshape_input`` :15-31,
``build_model`` :34-48, ``generate_random_hyperparameters`` :66-111); the network methods it calls
(``train_network``, ``test_network``, the checkpoint save) run on the HIP training / inference kernels.

The reference samples its training windows from a ZODB database (``ExampleDb.get_training_set``,
networks/trainingDB/ExampleDb.py:50-83), which is outside this path; the two databases here
expose the same ``get_training_set(size, ratio=2) -> (x_out, y_out, pos_count)`` over NPZ reads
(``raw`` + ``base_labels``, networks/reader.py:11-23) or synthetic squiggles, with the sampler's shape kept:
``size // ratio`` all-positive windows + the rest all-negative, shuffled, every window's labels uniform
(TrainingRead.get_pos / get_neg, networks/trainingDB/TrainingRead.py:226-257).
"""
from __future__ import annotations

import datetime
import os
import random

import numpy as np

from . import metrics
from .neural_network import build_model as _build_model


def reshape_input(data, window, n_inputs):
    """networks/train_validate.py:15-31: ``[-1, window, n_inputs]`` view of the data.  Data that does not fill whole
    windows is NOT an error there: two lengths are printed and the data comes back as it was; same here."""
    try:
        flat = np.asarray(data)
        fits = flat.size % (window * n_inputs) == 0
    except ValueError:                                     # ragged rows
        fits = False
    if fits:
        return flat.reshape(-1, window, n_inputs)
    for what in (data, data[0]):
        print(len(what))
    return data


def build_model(network_type, **kwargs):
    """networks/train_validate.py:34-48 (``save=True`` is passed through as a keyword, :323)."""
    return _build_model(network_type, saving=kwargs.pop("save", False), **kwargs)


def padding(data, window=35, n_input=1):
    """networks/train_validate.py:51-64: integer zeros up to the next multiple of the window -> (windows, how many).
    Unlike catfish/infer.py:32-36 an exact multiple gets NO extra window here."""
    tail = -len(data) % window
    if tail:
        data = np.hstack((data, np.zeros(tail, dtype=np.int64)))
    return reshape_input(data, window, n_input), tail


def generate_random_hyperparameters(network_type, learning_rate_min=-4, learning_rate_max=0,
                                    optimizer_list=("Adam", "RMSProp"), layer_size_list=(16, 32, 64, 128, 256),
                                    n_layers_min=1, n_layers_max=6, batch_size_list=(128, 256, 512),
                                    dropout_min=0.2, dropout_max=0.8, n_layers_res_min=1, n_layers_res_max=12,
                                    size_layers_res_list=(16, 32, 64, 128, 256)):
    """networks/train_validate.py:66-111: one random-search draw from numpy's GLOBAL generator, in the reference's
    call order (so a seeded run draws the same network): learning-rate exponent, optimizer, layer size, depth, batch
    size, keep probability, and for ResNetRNN a residual depth that is drawn and then DROPPED -- the reference stores
    ``n_layers`` under ``n_layers_res`` (:109) -- and the residual width.

    Every draw builds: the shipped geometry (``layer_size`` 64 / ``layer_size_res`` 32, any depth) runs on the tuned HIP
    kernels and trains natively; other sizes infer on the any-size HIP kernels (csrc/generic.hpp) and train through torch
    autograd with the recurrence on the same kernel family (anysize_train.py).
    """
    rs = np.random
    plan = [("learning_rate", lambda: float(10 ** rs.randint(learning_rate_min, learning_rate_max))),
            ("optimizer_choice", lambda: str(rs.choice(list(optimizer_list)))),
            ("layer_size", lambda: int(rs.choice(list(layer_size_list)))),
            ("n_layers", lambda: int(rs.randint(n_layers_min, n_layers_max))),
            ("batch_size", lambda: int(rs.choice(list(batch_size_list)))),
            ("keep_prob", lambda: round(rs.uniform(dropout_min, dropout_max), 1))]
    if network_type == "ResNetRNN":
        plan += [(None, lambda: rs.randint(n_layers_res_min, n_layers_res_max)),
                 ("layer_size_res", lambda: int(rs.choice(list(size_layers_res_list))))]
    drawn = {}
    for key, draw in plan:
        value = draw()
        if key is not None:
            drawn[key] = value
    if "layer_size_res" in drawn:
        drawn["n_layers_res"] = drawn["n_layers"]
    return drawn


# --------------------------------------------------------------------------- training databases
class WindowExampleDb(object):
    """Balanced sampler over two pools of uniform-label windows (ExampleDb.get_training_set's contract)."""

    def __init__(self, pos, neg, seed=None):
        self.pos = [np.asarray(p) for p in pos]
        self.neg = [np.asarray(n) for n in neg]
        self.nb_pos, self.nb_neg = len(self.pos), len(self.neg)
        self.rng = random.Random(seed)

    def get_training_set(self, size, ratio=2):
        """ExampleDb.py:50-83: ``size // ratio`` positives + the rest negatives, drawn without replacement per
        batch, shuffled; returns (tuple of windows, tuple of label lists, number of positive labels)."""
        nb_pos = size // ratio
        nb_neg = size - nb_pos
        ps = self.rng.sample(range(self.nb_pos), nb_pos)
        ns = self.rng.sample(range(self.nb_neg), nb_neg)
        data_out = [(self.pos[n], [1] * len(self.pos[n])) for n in ps] + [(self.neg[n], [0] * len(self.neg[n])) for n in ns]
        self.rng.shuffle(data_out)
        x_out, y_out = zip(*data_out)
        pos_count = sum(y.count(1) for y in y_out)
        return x_out, y_out, pos_count


def windows_from_labelled_read(raw, labels, width=34, lessen=1, max_neg=None, rng=None):
    """TrainingRead.get_pos / get_neg (TrainingRead.py:226-257): windows of ``width + 1`` samples centred on a point
    (one more sample right of the centre when the width is odd) whose labels are ALL 1 (positives, every
    ``lessen``-th hit) or ALL 0 (negatives, a random subset of ``max_neg`` centres)."""
    raw = np.asarray(raw)
    labels = np.asarray(labels).astype(np.int64)
    width_l = width // 2
    width_r = width - width_l
    n = len(labels)
    csum = np.concatenate(([0], np.cumsum(labels)))
    centres = np.arange(width_l, n - width_r)
    ones = csum[centres + width_r + 1] - csum[centres - width_l]          # label sum of [c - l, c + r]
    pos_c = centres[(labels[centres] == 1)][::lessen]
    pos_c = pos_c[(csum[pos_c + width_r + 1] - csum[pos_c - width_l]) == width + 1]
    neg_c = centres[(labels[centres] == 0) & (ones == 0)]
    if max_neg is not None and len(neg_c) > max_neg:
        rng = rng or np.random.default_rng(0)
        neg_c = np.sort(rng.choice(neg_c, size=max_neg, replace=False))
    pos = [raw[c - width_l:c + width_r + 1] for c in pos_c]
    neg = [raw[c - width_l:c + width_r + 1] for c in neg_c]
    return pos, neg


def load_npz(npz_file):
    """networks/reader.py:11-23: (raw signal, labels) of one NPZ read."""
    with np.load(npz_file, allow_pickle=False) as npz:
        return npz["raw"], npz["base_labels"]


def example_db_from_npz(npz_files, width=34, lessen=1, max_neg_per_read=2000, seed=0):
    """A training database over NPZ reads (raw = normalised signal, base_labels = 0/1 per sample)."""
    rng = np.random.default_rng(seed)
    pos, neg = [], []
    for f in npz_files:
        raw, labels = load_npz(f)
        p, n = windows_from_labelled_read(raw, labels, width, lessen, max_neg_per_read, rng)
        pos.extend(p)
        neg.extend(n)
    return WindowExampleDb(pos, neg, seed=seed)


def synthetic_labelled_read(length, seed, hp_fraction=0.06):
    """A normalised synthetic squiggle (SURVEY 8d generator) with a planted class: homopolymer stretches are long
    dwells at one level (what a homopolymer looks like in nanopore current), labelled 1."""
    rng = np.random.default_rng(seed)
    sig = np.empty(length, dtype=np.float64)
    lab = np.zeros(length, dtype=np.int64)
    i = 0
    while i < length:
        if rng.random() < hp_fraction / 8.0:
            n = int(rng.integers(40, 120))                 # homopolymer: one long flat event
            sig[i:i + n] = rng.normal(500.0, 60.0)
            lab[i:i + n] = 1
        else:
            n = int(rng.geometric(1.0 / 9.0))
            sig[i:i + n] = rng.normal(500.0, 60.0)
        i += n
    sig = np.clip(np.rint(sig + rng.normal(0.0, 8.0, size=length)), 0, 2047)
    shift = np.median(sig)
    scale = np.median(np.abs(sig - shift))
    return (sig - shift) / scale, lab


def synthetic_example_db(n_reads=8, read_len=20000, seed=0):
    pos, neg = [], []
    rng = np.random.default_rng(seed)
    for r in range(n_reads):
        raw, lab = synthetic_labelled_read(read_len, seed * 1000 + r)
        p, n = windows_from_labelled_read(raw, lab, 34, 1, 4000, rng)
        pos.extend(p)
        neg.extend(n)
    return WindowExampleDb(pos, neg, seed=seed)


# --------------------------------------------------------------------------- training loop
def _append(path, text):
    with open(path, "a+") as fh:
        fh.write(text)


def _checkpoint_round(network, step, batch_x, batch_y, report, validation):
    """What the reference does at a checkpoint step (networks/train_validate.py:154-175): save, score the batch just
    trained on, run one round of validation; the report lines go to ``report`` in that order."""
    network.save_network_to_model_path(step)
    saved = "Saved checkpoint at step {}\n".format(step)
    print(saved)
    _append(report, "\n" + saved)
    clock = datetime.datetime.now()
    # the reference feeds the TRAINING keep_prob here (:163), so its two numbers carry dropout noise; the
    # deterministic inference pass is reported instead
    batch_acc, batch_loss = network.evaluate(batch_x, batch_y)
    print("Validated in {}".format(datetime.datetime.now() - clock))
    _append(report, "\nTraining accuracy: {}\nTraining loss: {}\n".format(batch_acc, batch_loss))
    clock = datetime.datetime.now()
    _acc, precision, recall = validate(network, *validation)
    print("Validated in {}".format(datetime.datetime.now() - clock))
    _append(report, "Validation precision: {}\nValidation recall: {}\n".format(precision, recall))
    return batch_acc


def train_and_validate(network, db, training_nr, squiggles, max_seq_length, file_path, validation_start, max_number,
                       checkpoint_every=10000):
    """networks/train_validate.py:114-185.  ``training_nr // batch_size`` optimizer steps on balanced batches from
    ``db.get_training_set``; after the last step and after every ``checkpoint_every``-th (10 000 in the reference):
    checkpoint + batch metrics + one validation round.  Report text appended to ``file_path + ".txt"`` as the
    reference writes it.  Returns the last batch accuracy (None when no step ran; the reference fails on its unbound
    local there)."""
    report = file_path + ".txt"
    steps = training_nr // network.batch_size
    seen = steps * network.batch_size
    banner = "\nTraining on {} examples in {} batches\n".format(seen, steps)
    print("Start training at {}".format(datetime.datetime.now()))
    print(banner)
    _append(report, banner)
    validation = (squiggles, max_seq_length, file_path, validation_start, max_number)
    hp_labels = 0
    batch_acc = None
    for step in range(1, steps + 1):
        windows, labels, n_pos = db.get_training_set(network.batch_size, ratio=2)
        hp_labels += n_pos
        batch_x = reshape_input(windows, network.window, network.n_inputs)
        batch_y = reshape_input(labels, network.window, network.n_outputs)
        network.train_network(batch_x, batch_y, step)
        if step == steps or step % checkpoint_every == 0:
            if step == steps - 1:
                print("This was the final checkpoint\n")          # the reference's off-by-one message (:158)
            batch_acc = _checkpoint_round(network, step, batch_x, batch_y, report, validation)
    share = hp_labels / (network.window * seen) if seen else 0
    _append(report, "\nTraining set had {:.2%} HPs\n".format(share) + "\nFinished training!\n\n")
    return batch_acc


# --------------------------------------------------------------------------- validation: select, ONE packed launch, sums
_VALIDATION_REPORT = (
    "\n---NEXT ROUND OF VALIDATION---"
    "\nAverage performance of validation set:\n"
    "\tAccuracy: {mean_acc:.2%}\n"
    "\tLoss: {mean_loss:.4f}"
    "\nPerformance over whole set: \n"
    "\tDetected {tp} true positives, {fp} false positives, {tn} true negatives, {fn} false negatives in total.\n"
    "\tTrue number of HPs: {hp} \tTrue percentage: {hp_share:.2%}\t Predicted percentage HPs: {called_share:.2%}\n"
    "\tAccuracy: {acc:.2%}"
    "\n\tPrecision: {precision:.2%}\n\tRecall: {recall:.2%}"
    "\t\nF1 score: {f1:.4f}"
    "{closing}")
_VALIDATION_CLOSING = "\nFinished validation of model {} on {} raw signals of average length {}."


def select_validation_stretches(squiggles, window, max_seq_length, validation_start, max_number, loader=load_npz):
    """Which samples one validation round scores (networks/train_validate.py:214-249), in file order.

    ``validation_start``: "complete" = whole reads; an int = the stretch ``[start, start + n)``; "random" = a stretch
    at ``random.randint(0, len - n)`` (Python's global generator, one draw per read that is long enough, so a seeded
    run picks the reference's stretches); ``n`` = ``max_seq_length`` rounded down to whole windows.  Reads shorter
    than the stretch are skipped; the selection stops once ``max_number`` reads are in.
    -> (signals, labels): two lists of 1-D arrays."""
    whole = validation_start == "complete"
    if not whole and validation_start != "random" and type(validation_start) != int:
        raise ValueError("validation_start must be an int, 'random' or 'complete'")
    n = max_seq_length // window * window
    signals, labels = [], []
    for path in squiggles:
        raw, lab = loader(path)
        if not whole:
            room = len(raw) - n - (0 if validation_start == "random" else validation_start)
            if room < 0:
                continue
            first = random.randint(0, room) if validation_start == "random" else validation_start
            raw, lab = raw[first:first + n], lab[first:first + n]
        signals.append(np.asarray(raw))
        labels.append(np.asarray(lab))
        if len(signals) >= max_number:
            break
    return signals, labels


def pack_validation_windows(signals, labels, window):
    """All stretches as one window-major batch: x float32 [sum N_i, window, 1], y float64 [sum N_i * window], the first
    packed sample of every read (int64 [n_reads + 1]) and each read's zero tail.  The tail rule is the training
    driver's ``padding`` (:51-64): an exact multiple of the window gets none."""
    lengths = np.array([len(s) for s in signals], dtype=np.int64)
    n_win = -(-lengths // window)
    tails = n_win * window - lengths
    bounds = np.zeros(len(signals) + 1, dtype=np.int64)
    np.cumsum(n_win * window, out=bounds[1:])
    x = np.zeros(int(bounds[-1]), dtype=np.float32)
    y = np.zeros(int(bounds[-1]), dtype=np.float64)
    for b, s, l in zip(bounds[:-1].tolist(), signals, labels):
        x[b:b + len(s)] = s
        y[b:b + len(l)] = l
    return x.reshape(-1, window, 1), y, bounds, tails


def score_validation_batch(probs32, logits32, y, bounds, tails, threshold=0.5):
    """Per-read accuracy / loss and whole-batch confusion counts from ONE packed forward pass
    (what rnn_class.py:222-261 computes read by read):

    * accuracy_i = mean(round_half_even(p) == y) over the read's padded windows -- ``tf.round`` sends p = 0.5 to 0
      (rnn_class.py:85) -- as float32, an exact count divided once;
    * loss_i = mean sigmoid cross-entropy of the LOGITS (rnn_class.py:74-79), summed in double, float32;
    * counts: prediction = ``p >= threshold`` (so p = 0.5 counts as a call); a called sample is a true positive when
      its label is 1 and a false positive otherwise -- zero tails included --, an un-called one a true negative when
      its label is 0 and a false negative otherwise; then every tail sample is taken out of the true negatives whether
      or not it was one (rnn_class.py:245-249)."""
    p = np.asarray(probs32, dtype=np.float64).reshape(-1)
    z = np.asarray(logits32, dtype=np.float64).reshape(-1)
    called = p >= threshold
    counts = (int(np.count_nonzero(called & (y == 1))), int(np.count_nonzero(called & (y != 1))),
              int(np.count_nonzero(~called & (y == 0))) - int(tails.sum()), int(np.count_nonzero(~called & (y != 0))))
    right = np.concatenate(([0], np.cumsum(np.round(p) == y)))
    sizes = np.diff(bounds)
    with np.errstate(invalid="ignore", divide="ignore"):
        acc = (right[bounds[1:]] - right[bounds[:-1]]).astype(np.float32) / sizes.astype(np.float32)
        ce = np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z)))
        loss = np.array([np.sum(ce[a:b]) for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist())]) / sizes
    return acc, loss.astype(np.float32), counts


def validate(network, squiggles, max_seq_length, file_path, validation_start="random", max_number=856):
    """networks/train_validate.py:188-295 as one packed launch.

    The reference pushes every read through ``test_network`` (one ``sess.run`` each, up to 856 per round); windows are
    independent, so here the selected stretches of ALL reads form one ``[sum N_i, 35, 1]`` batch that goes through
    ``network.score_windows`` once, and the per-read numbers come out of array sums over the read boundaries.  Same
    report (appended to ``<basename of file_path>.txt`` in the current directory), prints, return value
    ``(accuracy, precision, recall)`` over the whole set, and the same bookkeeping: counts are added to the
    network's running ``tp/fp/tn/fn`` and those are reset afterwards.

    Kept behaviour of the loop being replaced: the read that reaches ``max_number`` is scored and counted but its
    accuracy / loss are left out of the averages, which still divide by the number of reads (:248-256); no read
    selected -> ZeroDivisionError."""
    print("Max length is {}".format(max_seq_length))
    print("Validation start is {}".format(validation_start))
    signals, labels = select_validation_stretches(squiggles, network.window, max_seq_length, validation_start,
                                                  max_number)
    n_reads = len(signals)
    n_samples = sum(len(s) for s in signals)
    if n_reads == 0:
        raise ZeroDivisionError("validation selected no read")
    x, y, bounds, tails = pack_validation_windows(signals, labels, network.window)
    probs, logits = network.score_windows(x)
    acc, loss, counts = score_validation_batch(probs, logits, y, bounds, tails)
    averaged = n_reads - 1 if n_reads >= max_number else n_reads
    acc_sum = loss_sum = 0
    for a, l in zip(acc[:averaged].tolist(), loss[:averaged].tolist()):      # doubles holding float32 values, in order
        acc_sum += a
        loss_sum += l
    network.tp, network.fp, network.tn, network.fn = (have + new for have, new in
                                                      zip((network.tp, network.fp, network.tn, network.fn), counts))
    tp, fp, tn, fn = network.tp, network.fp, network.tn, network.fn
    whole_acc = metrics.calculate_accuracy(tp, fp, tn, fn)
    precision, recall = metrics.precision_recall(tp, fp, fn)
    closing = _VALIDATION_CLOSING.format(network.model_type, n_reads, n_samples / n_reads)
    _append(file_path.split("/")[-1] + ".txt", _VALIDATION_REPORT.format(
        mean_acc=acc_sum / n_reads, mean_loss=loss_sum / n_reads, tp=tp, fp=fp, tn=tn, fn=fn, hp=tp + fn,
        hp_share=(tp + fn) / n_samples, called_share=(tp + fp) / n_samples, acc=whole_acc, precision=precision,
        recall=recall, f1=metrics.f1(precision, recall), closing=closing))
    print(closing)
    network.tp = network.fp = network.tn = network.fn = 0
    print("Validation accuracy: ", whole_acc)
    print("Validation loss: ", loss_sum / n_reads)
    return whole_acc, precision, recall


_USAGE = ("The following arguments should be provided in this order:\n"
          "\t-network type\n\t-path to training db"
          "\n\t-number of training reads\n\t-path to validation db"
          "\n\t-max length of validation reads\n\nOptional:"
          "\n\t-start position for validation\n\t-maximum number of reads for validation")
_SHIPPED_HPARAMS = {
    "RNN": dict(batch_size=256, optimizer_choice="Adam", learning_rate=0.001, layer_size=64, n_layers=3, keep_prob=0.8),
    "ResNetRNN": dict(batch_size=256, optimizer_choice="RMSProp", learning_rate=0.001, layer_size=64, n_layers=3,
                      keep_prob=0.8, layer_size_res=32, n_layers_res=2)}


def _npz_files(directory):
    return sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith(".npz"))


def main(argv):
    """networks/train_validate.py:298-360: ``network_type train_npz_dir n_training_examples validation_npz_dir
    max_validation_length [validation_start [max_number]]``.  The training "database" argument is a directory of NPZ
    reads (the reference takes a ZODB file); hyper-parameters are a random draw as in the reference (:328) unless
    CATFISH_SHIPPED_HPARAMS=1 asks for the shipped network's (the reference's commented block :329-332)."""
    args = list(argv[1:])
    if len(args) < 5:
        raise ValueError(_USAGE)
    kind, train_dir, val_dir = args[0], args[1], args[3]
    n_train, stretch = int(args[2]), int(args[4])
    start = int(args[5]) if len(args) > 5 else "random"
    most = int(args[6]) if len(args) > 6 else 856
    shipped = os.environ.get("CATFISH_SHIPPED_HPARAMS") == "1"
    hparams = dict(_SHIPPED_HPARAMS.get(kind, _SHIPPED_HPARAMS["ResNetRNN"])) if shipped \
        else generate_random_hyperparameters(kind)
    network = build_model(kind, save=True, **hparams)
    network.initialize_network()
    print("Loading training database..")
    db_train = example_db_from_npz(_npz_files(train_dir))
    print("Loading validation database..")
    squiggles = _npz_files(val_dir)
    began = datetime.datetime.now()
    train_and_validate(network, db_train, n_train, squiggles, stretch, network.model_path, start, most)
    print("Trained and validated network in {}".format(datetime.datetime.now() - began))
    print("Finished script at ", datetime.datetime.now())


if __name__ == "__main__":
    import sys
    main(sys.argv)
