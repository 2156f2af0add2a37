"""RNN -- the reference's model object (catfish/models/rnn_class.py:9-270) backed by the HIP engine.

Same constructor keywords, attributes and methods as the reference class; the
TensorFlow graph + ``tf.Session`` are replaced by one ``cf_model`` on one
MI355X (include/catfish_hip.h).  There is no CPU fallback: constructing the
engine without the built HIP library raises.
"""
from __future__ import annotations

import os

import numpy as np

from . import checkpoint
from . import metrics
from .engine import HipEngine, DEFAULT_MAX_WINDOWS


class RNN(object):
    def __init__(self, save=False, **kwargs):
        # adjustable parameters (rnn_class.py:13-19)
        self.batch_size = kwargs["batch_size"]
        self.optimizer_choice = kwargs["optimizer_choice"]
        self.learning_rate = kwargs["learning_rate"]
        self.layer_size = kwargs["layer_size"]
        self.n_layers = kwargs["n_layers"]
        self.keep_prob = kwargs["keep_prob"]
        self.keep_prob_test = 1.0

        # set parameters (rnn_class.py:25-32)
        self.n_inputs = 1
        self.n_outputs = 1
        self.window = 35
        self.layer_sizes = [self.layer_size, ] * self.n_layers
        self.saving_step = 10000
        self.cell_type = "GRU"
        if not hasattr(self, "_model_type"):
            self._model_type = "bi" + self.cell_type + "-RNN"

        # the reference validates the optimizer while building the graph (rnn_class.py:62-71)
        if self.optimizer_choice not in ("Adam", "RMSProp"):
            raise ValueError("Given optimizer choice is not known. Choose 'Adam' or 'RMSProp'.")

        # MI355X-specific knobs (not in the reference)
        self.device = int(kwargs.get("device", 0))
        self.max_windows_per_pass = int(kwargs.get("max_windows_per_pass", DEFAULT_MAX_WINDOWS))
        self.precision = kwargs.get("precision", "fp32")      # "fp32" (exact) | "bf16x3" | "bf16"

        self.weights = None
        self.engine = None
        self._trainer = None
        self._engine_stale = False
        self.train_loss = None
        self.train_seed = kwargs.get("train_seed")
        self._model_path = None
        if save:
            # rnn_class.py:43-46: claim a fresh model directory and write the model report; the TensorBoard
            # writer of the reference (:126-139) has no counterpart here (JSON logs instead)
            self.model_path = self.model_type
            self.save_info()

        # saving test performance (rnn_class.py:51-54)
        self.tp = 0
        self.fp = 0
        self.tn = 0
        self.fn = 0

    # ------------------------------------------------------------------ properties
    @property
    def model_type(self):
        return self._model_type

    @property
    def n_layers_res_(self):
        return 0

    @property
    def layer_size_res_(self):
        return 32

    @property
    def model_path(self):
        return self._model_path

    @model_path.setter
    def model_path(self, model_type):
        """rnn_class.py:100-118: the first free ``<dir>/<model_type>_<n>`` is created and becomes the model
        directory.  The reference hard-codes its authors' scratch directory (:102) and keeps ``os.getcwd()`` as the
        commented alternative (:103); here the parent is ``CATFISH_MODEL_DIR`` or the current directory."""
        cur_dir = os.environ.get("CATFISH_MODEL_DIR") or os.getcwd()
        number = 0
        while True:
            model_path = "{}/{}_{}".format(cur_dir, model_type, number)
            if not os.path.isdir(model_path):
                os.mkdir(model_path)
                break
            number += 1
        print("\nSaving network checkpoints to", model_path, "\n")
        self._model_path = model_path

    def use_model_path(self, model_path):
        """Adopt an existing model directory as ``model_path`` (no ``_<n>`` numbering), e.g. to continue a run."""
        os.makedirs(model_path, exist_ok=True)
        self._model_path = model_path

    # ------------------------------------------------------------------ weights
    def _load_engine(self, weights):
        if self.engine is not None:
            self.engine.close()
        self.weights = weights
        self.optimizer_state = None
        self._trainer = None
        self._engine_stale = False
        self.engine = HipEngine(weights, layer_size=self.layer_size, n_layers=self.n_layers,
                                layer_size_res=self.layer_size_res_, n_layers_res=self.n_layers_res_,
                                device=self.device, max_windows_per_pass=self.max_windows_per_pass,
                                precision=self.precision)

    def _initial_weights(self, seed=None):
        """TF default initialisers: glorot-uniform kernels, zero biases,
        GRU gates/bias = 1, gamma = 1, beta = 0, moving stats 0/1."""
        rng = np.random.default_rng(seed)
        w = {}

        def glorot(shape, fan_in, fan_out):
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, size=shape).astype(np.float32)

        c = self.layer_size_res_
        cin = 1
        for j in range(4 * self.n_layers_res_):
            k = 3 if j % 4 == 2 else 1
            c_in = cin if j % 4 in (0, 1) else c
            name = "conv1d" if j == 0 else "conv1d_%d" % j
            bn = "batch_normalization" if j == 0 else "batch_normalization_%d" % j
            w[name + "/kernel"] = glorot((k, c_in, c), k * c_in, k * c)
            w[name + "/bias"] = np.zeros(c, np.float32)
            w[bn + "/gamma"] = np.ones(c, np.float32)
            w[bn + "/beta"] = np.zeros(c, np.float32)
            w[bn + "/moving_mean"] = np.zeros(c, np.float32)
            w[bn + "/moving_variance"] = np.ones(c, np.float32)
            if j % 4 == 3:
                cin = c
        c_in = c if self.n_layers_res_ > 0 else 1
        h = self.layer_size
        for layer in range(self.n_layers):
            for d in ("fw", "bw"):
                p = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell" % (layer, d)
                w[p + "/gates/kernel"] = glorot((c_in + h, 2 * h), c_in + h, 2 * h)
                w[p + "/gates/bias"] = np.ones(2 * h, np.float32)
                w[p + "/candidate/kernel"] = glorot((c_in + h, h), c_in + h, h)
                w[p + "/candidate/bias"] = np.zeros(h, np.float32)
            c_in = 2 * h
        w["final_fully_connected/kernel"] = glorot((2 * h, 1), 2 * h, 1)
        w["final_fully_connected/bias"] = np.zeros(1, np.float32)
        return w

    def initialize_network(self, seed=None):
        """rnn_class.py:186-188: fresh variables (random init instead of a checkpoint)."""
        self._load_engine(self._initial_weights(seed))
        print("\nNot yet initialized: ", [], "\n")

    def restore_network(self, path, ckpnt="latest", meta=None):
        """rnn_class.py:191-198: load checkpoint ``path/ckpnt`` (or the latest one in ``path``)."""
        weights = checkpoint.read_inference_weights(path, ckpnt)
        self._load_engine(weights)
        # saver.restore also brings the optimizer slots back; train_network continues from them
        self.optimizer_state = checkpoint.read_optimizer_state(path, ckpnt)
        parts = path.split("/")
        print("Model {} restored\n".format(parts[-2] if len(parts) >= 2 else path))

    def set_weights(self, weights):
        """Load weights from a {TF variable name: array} dict (e.g. an exported .npz)."""
        self._load_engine(dict(weights))

    # ------------------------------------------------------------------ inference
    def _require_engine(self):
        if self.engine is None:
            raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
        if self._engine_stale:
            # weights moved by train_network: re-tile them into the HIP engine
            trainer = self._trainer
            self._load_engine(trainer.net.numpy_weights())
            self._trainer = trainer
            self._engine_stale = False

    def infer(self, input_x):
        """rnn_class.py:213-219: [N,35,1] -> float64 confidences, flattened [N*35]."""
        self._require_engine()
        try:
            import torch
            is_tensor = isinstance(input_x, torch.Tensor)
        except ImportError:
            is_tensor = False
        if is_tensor and input_x.is_cuda:
            out = self.engine.infer_device(input_x.to(dtype=torch.float32))
            return out.cpu().numpy().astype(float)
        confidences = self.engine.infer_host(np.asarray(input_x))
        return np.reshape(confidences, (-1)).astype(float)

    def score_windows(self, windows):
        """[N,35(,1)] windows -> (probabilities, logits), float32 [N*35] each, from ONE forward pass: ``self.predictions``
        (rnn_class.py:84) and the pre-sigmoid ``self.logits`` (rnn_class.py:178-183) the loss is defined on.  N is
        unbounded (the library walks it in passes), so a whole validation round is one call."""
        self._require_engine()
        return self.engine.infer_host(np.asarray(windows), return_logits=True)

    def test_network(self, test_x, test_y, read_name, file_path, padding_size, threshold=0.5):
        """rnn_class.py:222-261: accuracy and loss of one padded read + running confusion counters (the per-read
        form of the surface; ``train_validate.validate`` scores a whole round in one packed call instead)."""
        from .train_validate import score_validation_batch
        probs32, logits32 = self.score_windows(test_x)
        y = np.asarray(test_y, dtype=np.float64).reshape(-1)
        acc, loss, counts = score_validation_batch(probs32, logits32, y, np.array([0, y.size]),
                                                   np.array([padding_size]), threshold)
        self.tp, self.fp, self.tn, self.fn = (have + new for have, new in
                                              zip((self.tp, self.fp, self.tn, self.fn), counts))
        return float(acc[0]), float(loss[0])

    def evaluate(self, set_x, set_y):
        """``sess.run([self.accuracy, self.loss], ...)`` (networks/train_validate.py:162; rnn_class.py:74-88) on the
        inference graph: (accuracy, loss) of one batch without touching the confusion counters."""
        probs32, logits32 = self.score_windows(set_x)
        labels = np.asarray(set_y).reshape(-1)
        acc = float(np.mean(np.round(probs32.astype(float)) == labels))
        return acc, sigmoid_cross_entropy_from_logits(logits32.astype(np.float64), labels)

    def train_network(self, train_x, train_y, step):
        """rnn_class.py:201-210: one optimizer step on a batch (TF-1 update rules, dropout on the biGRU outputs).

        On a GPU the whole step -- forward, loss, backward, optimizer, weight re-tiling -- runs on HIP kernels through
        the C ABI (``training.Trainer`` -> ``native_step.py`` for the shipped 64 / 32 geometry, ``anysize_train.py``
        for other sizes); the torch restatement is the CPU mode and the test reference.  The inference engine picks
        the updated weights up lazily, at the next ``infer``.  TensorBoard summaries (the reference's second forward,
        :206-208) are not written.
        """
        if self.weights is None:
            raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
        if self._trainer is None:
            from .training import Trainer
            self._trainer = Trainer(self.weights, self.n_layers, self.n_layers_res_, self.optimizer_choice,
                                    self.learning_rate, self.keep_prob, seed=self.train_seed,
                                    optimizer_state=getattr(self, "optimizer_state", None))
        self.train_loss = self._trainer.train_step(train_x, train_y)
        self._engine_stale = True

    def save_network(self, path, step):
        """Write the current weights as a TensorFlow checkpoint-V2 bundle ``path/ckpnt-<step>`` that the
        original tool's ``restore_network`` (rnn_class.py:191-198) and this one can both read
        (the reference saves with ``saver.save(sess, ".../checkpoints/ckpnt", global_step=step)``,
        networks/train_validate.py:154-155).  Like tf.train.Saver it stores the optimizer slot variables
        (``<var>/RMSProp`` ... ) next to the weights once a training step has run."""
        if self.weights is None:
            raise RuntimeError("network has no weights: call restore_network() or initialize_network() first")
        weights = self._trainer.net.numpy_weights() if self._trainer is not None else self.weights
        prefix = os.path.join(path, "ckpnt-%d" % int(step))
        tensors = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        state = self._trainer.opt.state_tf() if self._trainer is not None else getattr(self, "optimizer_state", None)
        for k, v in (state or {}).items():
            tensors[k] = np.asarray(v, dtype=np.float32)
        checkpoint.write_checkpoint(prefix, tensors)
        with open(os.path.join(path, "checkpoint"), "w") as fh:
            fh.write('model_checkpoint_path: "ckpnt-%d"\nall_model_checkpoint_paths: "ckpnt-%d"\n' % (int(step), int(step)))
        return prefix

    def save_network_to_model_path(self, step):
        """networks/train_validate.py:154: ``saver.save(sess, model_path + "/checkpoints/ckpnt", global_step=step)``."""
        if self.model_path is None:
            raise RuntimeError("no model directory: construct the network with save=True or call use_model_path()")
        path = os.path.join(self.model_path, "checkpoints")
        os.makedirs(path, exist_ok=True)
        return self.save_network(path, step)

    def report_file(self):
        """Where ``save_info`` writes: the reference names it ``<model_path>.txt`` (NEXT TO the model directory,
        rnn_class.py:265); ``neural_network.load_network`` reads ``<network dir>/ResNetRNN.txt`` (neural_network.py:30),
        which is how the authors shipped it (catfish/ResNetRNN/ResNetRNN.txt).  Both are written."""
        return self.model_path + ".txt"

    def _write_report(self, text, mode):
        with open(self.report_file(), mode) as dest:
            dest.write(text)
        with open(os.path.join(self.model_path, "ResNetRNN.txt"), mode) as dest:
            dest.write(text)

    def save_info(self):
        """rnn_class.py:264-270: the model report header (``key: value`` lines retrieve_hyperparams parses)."""
        if self.model_path is None:
            raise RuntimeError("no model directory: construct the network with save=True or call use_model_path()")
        self._write_report("MODEL TYPE: {}\n\n".format(self.model_type)
                           + "batch_size: {}\noptimizer_choice: {}\nlearning_rate: {}\n".format(
                               self.batch_size, self.optimizer_choice, self.learning_rate)
                           + "layer_size: {}\nn_layers: {}\nkeep_prob: {}\n".format(
                               self.layer_size, self.n_layers, self.keep_prob), "w")


def sigmoid_cross_entropy_from_logits(logits, labels):
    """tf.losses.sigmoid_cross_entropy with default weights and reduction (rnn_class.py:74-79): the mean over all
    elements of ``max(z, 0) - z*y + log(1 + exp(-|z|))`` (TF's numerically stable form)."""
    z = np.asarray(logits, dtype=np.float64).reshape(-1)
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    return float(np.mean(np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z)))))
