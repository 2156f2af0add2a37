"""Model loader -- mirror of reference catfish/neural_network.py:8-67."""
from __future__ import annotations

from .resnet_class import ResNetRNN
from .rnn_class import RNN


def build_model(network_type, saving=False, **kwargs):
    """neural_network.py:8-23.  Like the reference, an unknown type is not rejected here
    (the reference then fails with UnboundLocalError at ``return network``)."""
    if network_type == "RNN":
        network = RNN(save=saving, **kwargs)
    elif network_type == "ResNetRNN":
        network = ResNetRNN(save=saving, **kwargs)
    return network  # noqa: F821  (UnboundLocalError for unknown types, as in the reference)


def load_network(network_type, path_to_network, checkpoint, **engine_kwargs):
    """neural_network.py:26-34: hyper-parameters from ``ResNetRNN.txt`` + checkpoint ``ckpnt-N``.

    ``engine_kwargs`` (device=, max_windows_per_pass=) are MI355X additions.
    """
    hpm_dict = retrieve_hyperparams(path_to_network + "/ResNetRNN.txt")
    hpm_dict.update(engine_kwargs)
    model = build_model(network_type, **hpm_dict)
    model.restore_network("{}/checkpoints".format(path_to_network), ckpnt="ckpnt-{}".format(checkpoint))
    return model


# (line prefix, key, cast) in the reference's match order; "layer_size:" / "n_layers:" carry the
# colon so that they do not swallow the *_res keys (neural_network.py:52-55 vs :58-61).
_HYPERPARAM_RULES = (
    ("batch_size", "batch_size", int),
    ("optimizer_choice", "optimizer_choice", str),
    ("learning_rate", "learning_rate", float),
    ("layer_size:", "layer_size", int),
    ("n_layers:", "n_layers", int),
    ("keep_prob", "keep_prob", float),
    ("layer_size_res", "layer_size_res", int),
    ("n_layers_res", "n_layers_res", int),
)


def retrieve_hyperparams(model_file, split_on=": "):
    """neural_network.py:37-67: the model report's ``key: value`` header lines, matched by prefix.

    Later lines override earlier ones; lines matching no rule (the validation log) are skipped.
    """
    found = {}
    with open(model_file, "r") as source:
        for line in source:
            for prefix, key, cast in _HYPERPARAM_RULES:
                if line.startswith(prefix):
                    found[key] = cast(line.strip().split(split_on)[1])
                    break
    return found
