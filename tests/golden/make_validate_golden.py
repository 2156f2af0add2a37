"""Generate the validation-report golden by EXECUTING the reference's own validation code.

The reference modules cannot be imported (``networks/train_validate.py`` and ``networks/rnn_class.py`` import
tensorflow, ``networks/trainingDB/metrics.py`` imports seaborn), so the definitions below are lifted out of the files
with ``ast`` and executed as they stand:

  networks/train_validate.py     : reshape_input (:15-31), padding (:51-64), validate (:188-295)
  networks/rnn_class.py          : RNN.test_network (:222-261)  -- bound to a stub object whose ``sess.run`` is
                                   tests/golden/validate_stub.py (a closed-form "network", see there)
  networks/trainingDB/metrics.py : confusion_matrix (:10-37), precision_recall (:40-55), calculate_accuracy (:58-63),
                                   f1 (:117-135)
  networks/reader.py             : load_npz (:11-23)

What is recorded per case: the report text ``validate`` appends to ``<basename>.txt``, everything it prints, its return
value, and the network's confusion counters afterwards.  The stub returns accuracy / loss as Python floats holding
float32 values: ``accuracy += sgl_acc`` then accumulates in double, which is what the reference's own numpy (1.x, where
``0 + np.float32`` is a float64) did; under numpy 2 a bare np.float32 would accumulate in float32 instead.

Outputs (data only): tests/golden/validate_golden.json + validate_golden_reads.npz.  Run in the build container
(needs /root/reference); the fixtures travel to the GPU box, the reference does not.
"""
import ast
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import validate_stub as stub                                                   # noqa: E402

REF = "/root/reference/networks"


def lift(path, names, ns, cls=None):
    """Execute the named top-level functions of ``path`` (or methods of class ``cls``) in namespace ``ns``."""
    with open(path) as fh:
        tree = ast.parse(fh.read())
    nodes = tree.body
    if cls is not None:
        nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    body = [n for n in nodes if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


class StubSession(object):
    """``tf.Session`` of the extracted test_network: the fetches are the tokens the stub network carries."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def run(self, fetches, feed_dict):
        logits = stub.stub_logits(feed_dict["x"], self.a, self.b)
        probs = stub.stub_probs(logits)
        if fetches == "predictions":
            return probs
        assert fetches == ["accuracy", "loss"]
        acc, loss = stub.stub_accuracy_loss(probs, logits, feed_dict["y"])
        return float(acc), float(loss)


def make_reads(rng):
    """Short labelled 'reads': raw = multiples of 0.25, labels = 0/1 runs; lengths around the window rules
    (exact multiples of 35, one off, shorter than a stretch)."""
    reads = []
    for n in (700, 735, 736, 1400, 300, 1051, 2100, 734, 35, 1225, 980, 1500):
        raw = np.clip(np.rint(np.cumsum(rng.normal(0, 0.6, size=n)) % 7.0 - 3.5 + rng.normal(0, 0.8, size=n)) / 4.0
                      + rng.integers(-4, 5, size=n) * 0.25, -3.0, 3.0)
        raw = np.round(raw * 4.0) / 4.0
        labels = np.zeros(n, dtype=np.int64)
        i = int(rng.integers(0, 60))
        while i < n:
            run = int(rng.integers(5, 45))
            labels[i:i + run] = 1
            i += run + int(rng.integers(20, 160))
        reads.append((raw, labels))
    return reads


CASES = [
    # name, validation_start, max_seq_length, max_number, (a, b), counters before, random seed
    ("fixed_start_all_reads", 35, 720, 856, (0.75, -0.375), (0, 0, 0, 0), None),
    ("fixed_start_early_break", 0, 700, 5, (0.75, -0.375), (0, 0, 0, 0), None),
    ("random_start", "random", 1000, 856, (1.5, -0.5), (0, 0, 0, 0), 7),
    ("random_start_early_break", "random", 735, 3, (1.5, -0.5), (0, 0, 0, 0), 11),
    ("complete_reads_padding_predicted_positive", "complete", 0, 856, (0.75, 0.375), (0, 0, 0, 0), None),
    ("complete_reads_counters_carried_in", "complete", 123, 7, (-1.0, 0.25), (3, 5, 7, 11), None),
    ("nothing_predicted_positive", 0, 350, 856, (0.0, -2.0), (0, 0, 0, 0), None),
    ("max_number_one", 100, 700, 1, (0.75, -0.375), (0, 0, 0, 0), None),
]


def main():
    tv = {"np": np, "os": os, "random": random}
    lift(os.path.join(REF, "train_validate.py"), {"reshape_input", "padding", "validate"}, tv)
    met = lift(os.path.join(REF, "trainingDB", "metrics.py"),
               {"confusion_matrix", "precision_recall", "calculate_accuracy", "f1"}, {"np": np})
    trainingDB = types.SimpleNamespace(metrics=types.SimpleNamespace(**{k: v for k, v in met.items() if callable(v)}))
    tv["trainingDB"] = trainingDB
    tv["reader"] = types.SimpleNamespace(**{"load_npz": lift(os.path.join(REF, "reader.py"), {"load_npz"},
                                                             {"np": np})["load_npz"]})
    rnn = lift(os.path.join(REF, "rnn_class.py"), {"test_network"}, {"np": np, "trainingDB": trainingDB}, cls="RNN")

    class StubNetwork(object):
        test_network = rnn["test_network"]
        x, y, p_dropout, predictions, accuracy, loss = "x", "y", "p_dropout", "predictions", "accuracy", "loss"
        window, n_inputs, n_outputs, keep_prob_test = 35, 1, 1, 1.0
        model_type = "ResNet-RNN"

    reads = make_reads(np.random.default_rng(20261004))
    out = {"cases": []}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for i, (raw, labels) in enumerate(reads):
            paths.append(os.path.join(tmp, "read_%02d.npz" % i))
            np.savez(paths[-1], raw=raw, base_labels=labels)
        os.chdir(tmp)
        try:
            for name, start, max_len, max_number, (a, b), before, seed in CASES:
                net = StubNetwork()
                net.sess = StubSession(a, b)
                net.tp, net.fp, net.tn, net.fn = before
                if seed is not None:
                    random.seed(seed)
                printed = io.StringIO()
                with contextlib.redirect_stdout(printed):
                    ret = tv["validate"](net, list(paths), max_len, "some/dir/" + name, start, max_number)
                with open(name + ".txt") as fh:
                    report = fh.read()
                out["cases"].append({
                    "name": name, "validation_start": start, "max_seq_length": max_len, "max_number": max_number,
                    "a": a, "b": b, "counters_before": list(before), "random_seed": seed, "report": report,
                    "printed": printed.getvalue(), "returned": [float(v) for v in ret],
                    "counters_after": [int(net.tp), int(net.fp), int(net.tn), int(net.fn)]})
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "validate_golden_reads.npz"),
                        **{"raw_%02d" % i: r for i, (r, _l) in enumerate(reads)},
                        **{"labels_%02d" % i: l for i, (_r, l) in enumerate(reads)})
    with open(os.path.join(HERE, "validate_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for c in out["cases"]:
        print(c["name"], c["returned"], c["counters_after"])
        print(c["report"])


if __name__ == "__main__":
    main()
