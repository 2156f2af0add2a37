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
    """networks/train_validate.py:15-31 (a failed reshape is printed and swallowed there; same here)."""
    try:
        data = np.reshape(data, (-1, window, n_inputs))
    except ValueError:
        print(len(data))
        print(len(data[0]))
    return data


def build_model(network_type, **kwargs):
    """networks/train_validate.py:34-48 (``save=True`` is passed through as a keyword, :323)."""
    return _build_model(network_type, saving=kwargs.pop("save", False), **kwargs)


def padding(data, window=35, n_input=1):
    """networks/train_validate.py:50-63: pad to a multiple of the window; unlike catfish/infer.py:32-36 an exact
    multiple gets NO extra window here (padding_size = 0)."""
    if not (len(data) / window).is_integer():
        padding_size = window - (len(data) - (len(data) // window * window))
        data = np.hstack((data, np.array(padding_size * [0])))
    else:
        padding_size = 0
    return reshape_input(data, window, n_input), padding_size


def generate_random_hyperparameters(network_type, learning_rate_min=-4, learning_rate_max=0,
                                    optimizer_list=("Adam", "RMSProp"), layer_size_list=(16, 32, 64, 128, 256),
                                    n_layers_min=1, n_layers_max=6, batch_size_list=(128, 256, 512),
                                    dropout_min=0.2, dropout_max=0.8, n_layers_res_min=1, n_layers_res_max=12,
                                    size_layers_res_list=(16, 32, 64, 128, 256)):
    """networks/train_validate.py:66-111, draw for draw (numpy's global generator, same call order).

    Kept quirk: the reference stores ``n_layers`` under ``n_layers_res`` (:109) and discards its own draw.
    Every draw builds: the shipped geometry (``layer_size`` 64 / ``layer_size_res`` 32, any depth) runs on the tuned HIP
    kernels and trains natively; other sizes infer on the any-size HIP kernels (csrc/generic.hpp) and train through torch
    autograd with the recurrence on the same kernel family (anysize_train.py).
    """
    learning_rate = 10 ** np.random.randint(learning_rate_min, learning_rate_max)
    optimizer = np.random.choice(list(optimizer_list))
    layer_size = np.random.choice(list(layer_size_list))
    n_layers = np.random.randint(n_layers_min, n_layers_max)
    batch_size = np.random.choice(list(batch_size_list))
    dropout = round(np.random.uniform(dropout_min, dropout_max), 1)
    hpm_dict = {"batch_size": int(batch_size), "optimizer_choice": str(optimizer), "learning_rate": float(learning_rate),
                "layer_size": int(layer_size), "n_layers": int(n_layers), "keep_prob": dropout}
    if network_type == "ResNetRNN":
        np.random.randint(n_layers_res_min, n_layers_res_max)          # drawn and dropped, as in the reference
        size_layers_res = np.random.choice(list(size_layers_res_list))
        hpm_dict["layer_size_res"] = int(size_layers_res)
        hpm_dict["n_layers_res"] = int(n_layers)
    return hpm_dict


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


# --------------------------------------------------------------------------- train / validate
def train_and_validate(network, db, training_nr, squiggles, max_seq_length, file_path, validation_start, max_number,
                       checkpoint_every=10000):
    """networks/train_validate.py:114-185: ``training_nr // batch_size`` optimizer steps on balanced batches; at the
    last step and every ``checkpoint_every`` (10 000 in the reference) steps: checkpoint, training metrics of the
    current batch, one round of validation.  Returns the last training accuracy."""
    print("Start training at {}".format(datetime.datetime.now()))
    n_examples = training_nr // network.batch_size * network.batch_size
    n_batches = n_examples // network.batch_size
    print("\nTraining on {} examples in {} batches\n".format(n_examples, n_batches))
    train_acc = None
    with open(file_path + ".txt", "a+") as dest:
        dest.write("\nTraining on {} examples in {} batches\n".format(n_examples, n_batches))
        step = 0
        positives = 0
        for _b in range(n_batches):
            data, labels, pos = db.get_training_set(network.batch_size, ratio=2)
            positives += pos
            set_x = reshape_input(data, network.window, network.n_inputs)
            set_y = reshape_input(labels, network.window, network.n_outputs)
            step += 1                                                           # step is per batch
            network.train_network(set_x, set_y, step)
            if step == n_batches or step % checkpoint_every == 0:
                network.save_network_to_model_path(step)
                print("Saved checkpoint at step {}\n".format(step))
                dest.write("\nSaved checkpoint at step {}\n".format(step))
                # training performance on the current batch (the reference evaluates accuracy and loss with the
                # training keep_prob fed, :162; here the deterministic inference graph is used)
                t1 = datetime.datetime.now()
                train_acc, train_loss = network.evaluate(set_x, set_y)
                print("Validated in {}".format(datetime.datetime.now() - t1))
                dest.write("\nTraining accuracy: {}\n".format(train_acc))
                dest.write("Training loss: {}\n".format(train_loss))
                t1 = datetime.datetime.now()
                dest.flush()
                _val_acc, whole_precision, whole_recall = validate(network, squiggles, max_seq_length, file_path,
                                                                   validation_start, max_number)
                print("Validated in {}".format(datetime.datetime.now() - t1))
                dest.write("Validation precision: {}\n".format(whole_precision))
                dest.write("Validation recall: {}\n".format(whole_recall))
        try:
            train_hp = positives / (network.window * n_examples)
        except ZeroDivisionError:
            train_hp = 0
        dest.write("\nTraining set had {:.2%} HPs\n".format(train_hp))
        dest.write("\nFinished training!\n\n")
    return train_acc


def validate(network, squiggles, max_seq_length, file_path, validation_start="random", max_number=856):
    """networks/train_validate.py:188-295: per NPZ read a stretch of ``max_seq_length`` samples (fixed start, random
    start, or the "complete" read) through ``network.test_network``; report text appended to ``<basename>.txt`` in the
    current directory exactly as the reference writes it.  Returns (accuracy, precision, recall) over the whole set
    and resets the network's confusion counters."""
    total_length = 0
    accuracy = 0
    loss = 0
    valid_reads = 0
    print("Max length is {}".format(max_seq_length))
    print("Validation start is {}".format(validation_start))
    file_path = file_path.split("/")[-1]
    for squig in squiggles:
        data_sq, labels_sq = load_npz(squig)
        if validation_start == "complete":
            total_length += len(data_sq)
        else:
            max_seq_length = max_seq_length // network.window * network.window
            if type(validation_start) == int:
                if len(data_sq) >= validation_start + max_seq_length:
                    start_val = validation_start
                else:
                    continue
            elif validation_start == "random":
                if len(data_sq) >= max_seq_length:
                    start_val = random.randint(0, len(data_sq) - max_seq_length)
                else:
                    continue
            labels_sq = labels_sq[start_val: start_val + max_seq_length]
            data_sq = data_sq[start_val: start_val + max_seq_length]
            total_length += max_seq_length
        read_name = os.path.basename(squig).split(".npz")[0]
        valid_reads += 1
        set_x, padding_size = padding(data_sq, network.window, network.n_inputs)
        set_y, _ = padding(labels_sq, network.window, network.n_inputs)
        sgl_acc, sgl_loss = network.test_network(set_x, set_y, read_name, file_path, padding_size)
        if valid_reads >= max_number:
            break                                   # kept: the last read's accuracy / loss are not added (:253-260)
        accuracy += sgl_acc
        loss += sgl_loss

    whole_accuracy = metrics.calculate_accuracy(network.tp, network.fp, network.tn, network.fn)
    whole_precision, whole_recall = metrics.precision_recall(network.tp, network.fp, network.fn)
    whole_f1 = metrics.f1(whole_precision, whole_recall)
    with open(file_path + ".txt", "a+") as dest:
        dest.write("\n---NEXT ROUND OF VALIDATION---")
        dest.write("\nAverage performance of validation set:\n")
        dest.write("\tAccuracy: {:.2%}\n".format(accuracy / valid_reads))
        dest.write("\tLoss: {0:.4f}".format(loss / valid_reads))
        dest.write("\nPerformance over whole set: \n")
        dest.write("\tDetected {} true positives, {} false positives, {} true negatives, {} false negatives in total.\n".
                   format(network.tp, network.fp, network.tn, network.fn))
        dest.write("\tTrue number of HPs: {} \tTrue percentage: {:.2%}\t Predicted percentage HPs: {:.2%}\n".
                   format(network.tp + network.fn, (network.tp + network.fn) / (total_length),
                          (network.tp + network.fp) / (total_length)))
        dest.write("\tAccuracy: {:.2%}".format(whole_accuracy))
        dest.write("\n\tPrecision: {:.2%}\n\tRecall: {:.2%}".format(whole_precision, whole_recall))
        dest.write("\t\nF1 score: {0:.4f}".format(whole_f1))
        dest.write("\nFinished validation of model {} on {} raw signals of average length {}.".format(
            network.model_type, valid_reads, total_length / valid_reads))
    print("\nFinished validation of model {} on {} raw signals of average length {}.".format(
        network.model_type, valid_reads, total_length / valid_reads))
    network.tp = 0
    network.fn = 0
    network.tn = 0
    network.fp = 0
    print("Validation accuracy: ", whole_accuracy)
    print("Validation loss: ", loss / valid_reads)
    return whole_accuracy, whole_precision, whole_recall


def main(argv):
    """networks/train_validate.py:298-360: ``network_type train_npz_dir n_training_examples validation_npz_dir
    max_validation_length [validation_start [max_number]]``.  The training "database" argument is a directory of NPZ
    reads (the reference takes a ZODB file); hyper-parameters are a random draw as in the reference (:328) unless
    CATFISH_SHIPPED_HPARAMS=1 asks for the shipped network's (the reference's commented block :329-332)."""
    if len(argv) < 6:
        raise ValueError("The following arguments should be provided in this order:\n" +
                         "\t-network type\n\t-path to training db" +
                         "\n\t-number of training reads\n\t-path to validation db" +
                         "\n\t-max length of validation reads\n\nOptional:" +
                         "\n\t-start position for validation\n\t-maximum number of reads for validation")
    network_type = argv[1]
    db_dir_train = argv[2]
    training_nr = int(argv[3])
    db_dir_val = argv[4]
    max_seq_length = int(argv[5])
    validation_start = int(argv[6]) if len(argv) >= 7 else "random"
    max_number = int(argv[7]) if len(argv) >= 8 else 856
    if os.environ.get("CATFISH_SHIPPED_HPARAMS") != "1":
        hpm_dict = generate_random_hyperparameters(network_type)
    elif network_type == "RNN":
        hpm_dict = {"batch_size": 256, "optimizer_choice": "Adam", "learning_rate": 0.001, "layer_size": 64,
                    "n_layers": 3, "keep_prob": 0.8}
    else:
        hpm_dict = {"batch_size": 256, "optimizer_choice": "RMSProp", "learning_rate": 0.001, "layer_size": 64,
                    "n_layers": 3, "keep_prob": 0.8, "layer_size_res": 32, "n_layers_res": 2}
    network = build_model(network_type, save=True, **hpm_dict)
    network.initialize_network()
    print("Loading training database..")
    db_train = example_db_from_npz(sorted(os.path.join(db_dir_train, f) for f in os.listdir(db_dir_train)
                                          if f.endswith(".npz")))
    print("Loading validation database..")
    squiggles = sorted(os.path.join(db_dir_val, f) for f in os.listdir(db_dir_val) if f.endswith(".npz"))
    t5 = datetime.datetime.now()
    train_and_validate(network, db_train, training_nr, squiggles, max_seq_length, network.model_path,
                       validation_start, max_number)
    print("Trained and validated network in {}".format(datetime.datetime.now() - t5))
    print("Finished script at ", datetime.datetime.now())


if __name__ == "__main__":
    import sys
    main(sys.argv)
