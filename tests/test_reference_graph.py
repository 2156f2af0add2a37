"""The oracle (and the torch training restatement) against the reference's OWN TensorFlow graph.

``tests/golden/graph_golden.npz`` holds the outputs of the MetaGraphDef the reference ships
(ckpnt-30000.meta), interpreted node by node in numpy (``oracle/tf_graph.py``,
generator ``tests/golden/make_graph_golden.py``).  The graph's structure and attributes are the
reference's; only the per-op arithmetic is restated.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, has_reference
from oracle import catfish_oracle as oracle


@pytest.fixture(scope="module")
def graph_summary():
    with open(os.path.join(GOLDEN, "graph_summary.json")) as fh:
        return json.load(fh)


def test_oracle_matches_reference_graph_fp64(ckpt_weights, graph_golden):
    """Exact-arithmetic value of the reference graph vs the fp64 oracle (bundled checkpoint).

    The only known difference is the BN epsilon: the graph holds float32(1e-3) = 0.0010000000475, the
    oracle the double 1e-3 -> ~1e-10 on the probabilities.
    """
    p, st = oracle.forward(graph_golden["x"], ckpt_weights, np.float64, return_stages=True)
    assert np.abs(p - graph_golden["ckpt_f64_probs"]).max() < 1e-9
    assert np.abs(st["logits"] - graph_golden["ckpt_f64_logits"].reshape(-1)).max() < 1e-7
    n = graph_golden["ckpt_f64_res0"].shape[0]
    for k in ("res0", "res1", "gru0", "gru1", "gru2"):
        ref = graph_golden["ckpt_f64_" + k]
        assert st[k][:n].shape == ref.shape, k
        assert np.allclose(st[k][:n], ref, rtol=2e-8, atol=1e-8), k


def test_oracle_matches_reference_graph_fp32(ckpt_weights, graph_golden):
    """The graph run in float32 (as TF does) vs the fp32 oracle: rounding-order differences only."""
    p = oracle.forward(graph_golden["x"], ckpt_weights, np.float32)
    assert np.abs(p - graph_golden["ckpt_f32_probs"]).max() < 5e-6
    assert np.abs(graph_golden["ckpt_f32_probs"] - graph_golden["ckpt_f64_probs"]).max() < 5e-6


def test_oracle_matches_reference_graph_random_weights(graph_golden):
    """Same graph, seeded random variables: a structure check that does not depend on the trained values."""
    w = oracle.random_weights(seed=int(graph_golden["random_seed"]))
    p = oracle.forward(graph_golden["x"], w, np.float64)
    assert np.abs(p - graph_golden["rand_f64_probs"]).max() < 1e-9


def test_graph_summary_matches_what_the_kernels_assume(graph_summary):
    """Attributes the HIP kernels hard-code, as they stand in the reference's graph."""
    assert graph_summary["meta_info"]["tensorflow_version"] == "1.10.0"
    assert len(graph_summary["variables_inference"]) == 74
    assert len(graph_summary["conv2d_attrs"]) == 8
    for name, at in graph_summary["conv2d_attrs"].items():
        assert at["padding"] == "SAME" and at["strides"] == [1, 1, 1, 1] and at["data_format"] == "NHWC", name
    eps = [v for k, v in graph_summary["constants"].items() if k.endswith("batchnorm/add/y")]
    assert len(eps) == 8 and all(abs(e - 1e-3) < 1e-9 for e in eps)
    assert set(graph_summary["split_num"].values()) == {2} and len(graph_summary["split_num"]) == 6
    ops = graph_summary["ops_inference"]
    assert ops["Exit"] == 6 and ops["ReverseV2"] == 6 and ops["MatMul"] == 13 and ops["Sigmoid"] == 7 and ops["Tanh"] == 6
    opt = graph_summary["optimizer_constants"]
    assert abs(opt["optimizer/RMSProp/learning_rate"] - 1e-3) < 1e-9
    assert abs(opt["optimizer/RMSProp/decay"] - 0.9) < 1e-7 and opt["optimizer/RMSProp/momentum"] == 0.0
    assert abs(opt["optimizer/RMSProp/epsilon"] - 1e-10) < 1e-16


def test_training_loss_and_accuracy_match_reference_graph(ckpt_weights, graph_golden):
    """loss/Mean and accuracy/Mean of the reference graph vs catfish_amd.training / catfish_amd.metrics."""
    import torch
    from catfish_amd.training import TorchResNetRNN
    net = TorchResNetRNN(ckpt_weights, 3, 2, device="cpu", dtype=torch.float64)
    x = torch.from_numpy(graph_golden["x"]).to(torch.float64)
    y = torch.from_numpy(graph_golden["y"].reshape(-1, 35)).to(torch.float64)
    loss = float(net.loss(x, y))
    assert abs(loss - float(graph_golden["ckpt_f64_loss"])) < 1e-9
    p = oracle.forward(graph_golden["x"], ckpt_weights, np.float64)
    acc = np.mean(np.round(p) == graph_golden["y"].reshape(-1))
    assert abs(acc - float(graph_golden["ckpt_f64_accuracy"])) < 1e-6


@pytest.mark.skipif(not has_reference(), reason="needs /root/reference (build container only)")
def test_golden_regenerates_from_the_reference_meta_graph(graph_golden):
    """Re-interpret the reference's .meta live and compare with the committed vectors."""
    from catfish_amd import checkpoint
    from oracle import tf_graph
    prefix = os.path.join(REFERENCE, "catfish", "ResNetRNN", "checkpoints", "ckpnt-30000")
    nodes = tf_graph.load_meta_graph(prefix + ".meta")
    gi = tf_graph.GraphInterpreter(nodes, checkpoint.read_checkpoint(prefix), np.float64)
    x = graph_golden["x"][:12]
    p = gi.run("accuracy/Sigmoid", {"data/Placeholder": x.reshape(-1, 35, 1), "dropout": np.float32(1.0)})
    assert np.array_equal(np.asarray(p).reshape(-1), graph_golden["ckpt_f64_probs"][:12 * 35])
    # dropout wiring: keep_prob < 1 must change the result (the cells are wrapped, rnn_class.py:152)
    gi2 = tf_graph.GraphInterpreter(nodes, checkpoint.read_checkpoint(prefix), np.float64, seed=1)
    q = gi2.run("accuracy/Sigmoid", {"data/Placeholder": x.reshape(-1, 35, 1), "dropout": np.float32(0.8)})
    assert np.abs(np.asarray(q).reshape(-1) - np.asarray(p).reshape(-1)).max() > 1e-4


# ------------------------------------------------------------------------------------------------
# The training step (rnn_class.py:62-71, 201-210): the reference's gradient + ApplyRMSProp subgraph
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def train_golden():
    with np.load(os.path.join(GOLDEN, "graph_train_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def _close(a, b, rtol):
    return np.abs(a - b).max() <= rtol * np.abs(b).max() + 1e-12


def test_training_trajectory_matches_reference_graph(ckpt_weights, train_golden):
    """Three optimizer steps of the reference's own graph (loss -> optimizer/gradients/* -> ApplyRMSProp,
    interpreted in fp64) vs catfish_amd.training in fp64: losses, first-step gradients of all 58 trainable
    variables, and the accumulated weight change."""
    import torch
    from catfish_amd.training import Trainer
    tr = Trainer(ckpt_weights, 3, 2, "RMSProp", 1e-3, 1.0, device="cpu", use_graph=False, native=False, dtype=torch.float64)
    xs, ys = train_golden["train_x"], train_golden["train_y"]
    start = {k: v.detach().clone() for k, v in tr.net.trainable().items()}
    assert len(start) == 58
    # first-step gradients
    loss = tr.net.loss(torch.from_numpy(xs[0]).double(), torch.from_numpy(ys[0]).double())
    loss.backward()
    for k, p in tr.net.trainable().items():
        assert _close(p.grad.numpy(), train_golden["train_grad0/" + k], 1e-6), k
        p.grad = None
    for step in range(xs.shape[0]):
        got = tr.train_step(xs[step], ys[step])
        assert abs(got - float(train_golden["train_loss"][step])) < 1e-9, step
    for k, p in tr.net.trainable().items():
        delta = (p.detach() - start[k]).numpy()
        assert _close(delta, train_golden["train_delta/" + k], 1e-6), k


@pytest.mark.skipif(not has_reference(), reason="needs /root/reference (build container only)")
def test_restore_then_train_continues_from_the_saved_slots(tmp_path):
    """saver.restore brings the RMSProp slots back (rnn_class.py:191-198); one step from ckpnt-30000's own slot
    variables, reference graph vs TFOptimizer.load_state_tf, and the slots survive a save/restore round trip."""
    import torch
    from catfish_amd import checkpoint
    from catfish_amd.training import Trainer
    from oracle import tf_graph
    path = os.path.join(REFERENCE, "catfish", "ResNetRNN", "checkpoints")
    variables = checkpoint.read_checkpoint(os.path.join(path, "ckpnt-30000"))
    state = checkpoint.read_optimizer_state(path, "ckpnt-30000")
    weights = checkpoint.read_inference_weights(path, "ckpnt-30000")
    assert len(state) == 116 and len(weights) == 74
    nodes = tf_graph.load_meta_graph(os.path.join(path, "ckpnt-30000.meta"))
    rng = np.random.RandomState(5)
    x = (rng.randn(8, 35) * 1.5).astype(np.float32)
    y = (rng.rand(8, 35) < 0.3).astype(np.float32)
    gi = tf_graph.GraphInterpreter(nodes, variables, np.float64)
    applies = sorted(k for k, n in nodes.items() if n.op == "ApplyRMSProp")
    gi.run(applies, {"data/Placeholder": x.reshape(-1, 35, 1), "data/Placeholder_1": y.reshape(-1, 35, 1), "dropout": np.float32(1.0)})
    tr = Trainer(weights, 3, 2, "RMSProp", 1e-3, 1.0, device="cpu", use_graph=False, native=False, dtype=torch.float64,
                 optimizer_state=state)
    tr.train_step(x, y)
    for k, p in tr.net.trainable().items():
        ref_delta = gi.updates[k] - variables[k].astype(np.float64)
        # 1e-6: the graph's constants are float32 (lr 0.0010000000475, decay 0.89999998), the trainer's are doubles
        assert _close(p.detach().numpy() - variables[k].astype(np.float64), ref_delta, 1e-6), k
        assert _close(tr.opt.ms[k].numpy(), gi.updates[k + "/RMSProp"], 1e-6), k
        assert _close(tr.opt.mom[k].numpy(), gi.updates[k + "/RMSProp_1"], 1e-6), k
    # the slots travel through save_network's bundle
    tensors = dict(tr.net.numpy_weights())
    tensors.update(tr.opt.state_tf())
    checkpoint.write_checkpoint(str(tmp_path / "ckpnt-1"), tensors)
    back = checkpoint.read_optimizer_state(str(tmp_path), "ckpnt-1")
    assert set(back) == set(state)
    assert np.array_equal(back["conv1d/kernel/RMSProp"], tr.opt.ms["conv1d/kernel"].numpy().astype(np.float32))


def test_dropout_wiring_matches_reference_graph(ckpt_weights, train_golden):
    """keep_prob 0.8 (rnn_class.py:146-152): replaying the masks the reference graph drew, the restated
    training graph gives the same loss and the same gradients -- dropout sits on each cell's OUTPUT only (the
    carried state is not dropped), scaled by 1/keep_prob, and the mask multiplies the backward signal."""
    import json
    import torch
    from catfish_amd.training import TorchResNetRNN
    g = train_golden
    masks = {(layer, d): g["drop_masks"][layer, di] for layer in range(3) for di, d in enumerate(("fw", "bw"))}
    frac = float(g["drop_masks"].mean())
    assert 0.78 < frac < 0.82                                   # keep_prob 0.8
    net = TorchResNetRNN(ckpt_weights, 3, 2, device="cpu", dtype=torch.float64)
    loss = net.loss(torch.from_numpy(g["drop_x"]).double(), torch.from_numpy(g["drop_y"]).double(),
                    keep_prob=float(g["drop_keep_prob"]), masks=masks)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["drop_loss"])) < 1e-9
    sums = json.loads(str(g["drop_grad_sums_json"]))
    for k, p in net.trainable().items():
        gr = p.grad.numpy()
        scale = sums[k][1] + 1e-30
        assert abs(gr.sum() - sums[k][0]) < 1e-7 * scale and abs(np.abs(gr).sum() - sums[k][1]) < 1e-7 * scale, k
        if "drop_grad/" + k in g:
            assert _close(gr, g["drop_grad/" + k], 1e-6), k
    # without the masks (keep_prob 1) the loss is different: the test would notice a dropped wrapper
    plain = float(net.loss(torch.from_numpy(g["drop_x"]).double(), torch.from_numpy(g["drop_y"]).double()).detach())
    assert abs(plain - float(g["drop_loss"])) > 1e-4
