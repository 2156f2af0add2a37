"""ResNetRNN -- reference catfish/models/resnet_class.py:7-42 on the HIP engine."""
from __future__ import annotations

from .rnn_class import RNN


class ResNetRNN(RNN):
    def __init__(self, **kwargs):
        self.n_layers_res = kwargs["n_layers_res"]
        self.layer_size_res = kwargs["layer_size_res"]
        self.network_type = "ResNet-RNN"
        self._model_type = self.network_type
        RNN.__init__(self, **kwargs)

    def save_info(self):
        """resnet_class.py:28-32: the RNN header plus the two residual-stack lines."""
        RNN.save_info(self)
        self._write_report("layer_size_res: {}\nn_layers_res: {}\n\n".format(self.layer_size_res, self.n_layers_res), "a")

    @property
    def model_type(self):
        return self._model_type

    @property
    def n_layers_res_(self):
        return self.n_layers_res

    @property
    def layer_size_res_(self):
        return self.layer_size_res
