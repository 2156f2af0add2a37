"""Golden vectors from the reference's OWN TensorFlow graph.

Reads ``catfish/ResNetRNN/checkpoints/ckpnt-30000.meta`` (the MetaGraphDef the reference's
TF-1.10 session ran, saved at rnn_class.py:48 / restored at rnn_class.py:191-198) with the
wire-format parser in ``oracle/tf_graph.py`` -- nothing from the file is executed -- and
evaluates its inference subgraph ``data/Placeholder -> accuracy/Sigmoid`` (rnn_class.py:84,
213-216), the loss ``loss/Mean`` (rnn_class.py:74-79) and ``accuracy/Mean`` node by node in numpy,
with the variables of ``ckpnt-30000`` and with a seeded random weight set.

Run in the build container (needs /root/reference):   python tests/golden/make_graph_golden.py
Writes tests/golden/graph_golden.npz, tests/golden/graph_train_golden.npz (three optimizer steps of the
reference's gradient + ApplyRMSProp subgraph, rnn_class.py:62-71,201-210) and tests/golden/graph_summary.json.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from catfish_amd import checkpoint            # noqa: E402  (bundle reader, pinned by the bundle's own CRC-32Cs)
from oracle import catfish_oracle as O        # noqa: E402  (only for the seeded random weight set)
from oracle import tf_graph as G              # noqa: E402

REF = os.environ.get("CATFISH_REFERENCE", "/root/reference")
PREFIX = os.path.join(REF, "catfish", "ResNetRNN", "checkpoints", "ckpnt-30000")
FETCH = "accuracy/Sigmoid"
STAGES = {
    "res0": "ResNet_layer_0/residual_block/Relu_3",
    "res1": "ResNet_layer_1/residual_block/Relu_3",
    "gru0": "recurrent_layer/stack_bidirectional_rnn/cell_0/concat",
    "gru1": "recurrent_layer/stack_bidirectional_rnn/cell_1/concat",
    "gru2": "recurrent_layer/stack_bidirectional_rnn/cell_2/concat",
    "logits": "Reshape_1",
}
RANDOM_SEED = 3
STAGE_WINDOWS = 8


def make_inputs():
    rng = np.random.RandomState(20260101)
    x = (rng.randn(96, 35) * 1.5).astype(np.float32)
    x[5] = 0.0                                  # an all-padding window
    x[6, 20:] = 0.0                             # a ragged tail (infer.py:31-38 zero padding)
    x[7] *= 8.0                                 # saturating activations
    y = (rng.rand(96, 35, 1) < 0.3).astype(np.float32)
    return x, y


def run(nodes, variables, x, y, dtype):
    gi = G.GraphInterpreter(nodes, variables, dtype)
    feeds = {"data/Placeholder": x.reshape(-1, 35, 1), "data/Placeholder_1": y, "dropout": np.float32(1.0)}
    out = {"probs": np.asarray(gi.run(FETCH, feeds)).reshape(-1)}
    for k, node in STAGES.items():
        out[k] = np.asarray(gi.run(node, feeds))
    out["loss"] = np.asarray(gi.run("loss/Mean", feeds))
    out["accuracy"] = np.asarray(gi.run("accuracy/Mean", feeds))
    return out, gi.ops_used


TRAIN_STEPS = 3
TRAIN_BATCH = 16


def train_trajectory(nodes, variables, store32=False, tag="train"):
    """TRAIN_STEPS session.run(optimizer) calls of the reference graph (rnn_class.py:201-210) in exact arithmetic:
    loss -> optimizer/gradients/* -> ApplyRMSProp, from the checkpoint's weights with FRESH slot variables
    (rms = 1, momentum = 0: the optimizer's own initial values), dropout keep_prob = 1.

    store32: round every assigned variable to float32 between steps, as TF's float32 variables do (the arithmetic
    of each step stays exact).  A float32 trainer follows THAT trajectory; the exact one is for float64 runs: the
    one-ulp storage rounding of step k moves the gradients of step k+1 by up to ~1 %."""
    rng = np.random.RandomState(20260102)
    xs = (rng.randn(TRAIN_STEPS, TRAIN_BATCH, 35) * 1.5).astype(np.float32)
    ys = (rng.rand(TRAIN_STEPS, TRAIN_BATCH, 35) < 0.3).astype(np.float32)
    state = {k: np.asarray(v, dtype=np.float64) for k, v in variables.items()}
    for k in state:
        if k.endswith("/RMSProp"):
            state[k] = np.ones_like(state[k])
        elif k.endswith("/RMSProp_1"):
            state[k] = np.zeros_like(state[k])
    start = {k: v.copy() for k, v in state.items()}
    applies = sorted(k for k, n in nodes.items() if n.op == "ApplyRMSProp")
    grad_of = {nodes[a].inputs[0][0]: "%s:%d" % nodes[a].inputs[7] for a in applies}
    out = {"train_x": xs, "train_y": ys, tag + "_loss": np.zeros(TRAIN_STEPS)}
    for step in range(TRAIN_STEPS):
        gi = G.GraphInterpreter(nodes, state, np.float64)
        names = sorted(grad_of)
        res = gi.run(["loss/Mean"] + [grad_of[k] for k in names] + applies,
                     {"data/Placeholder": xs[step].reshape(-1, 35, 1), "data/Placeholder_1": ys[step].reshape(-1, 35, 1),
                      "dropout": np.float32(1.0)})
        out[tag + "_loss"][step] = float(res[0])
        if step == 0 and not store32:
            for k, g in zip(names, res[1:1 + len(names)]):
                out["train_grad0/" + k] = np.asarray(g).reshape(state[k].shape).astype(np.float32)
        assert len(gi.updates) == 3 * len(applies)
        if store32:
            state.update({k: v.astype(np.float32).astype(np.float64) for k, v in gi.updates.items()})
        else:
            state.update(gi.updates)
    for k in sorted(grad_of):
        out[tag + "_delta/" + k] = (state[k] - start[k]).astype(np.float32)
    return out


DROP_KEEP = 0.8          # the shipped ResNetRNN.txt keep_prob
DROP_BATCH = 8


def dropout_step(nodes, variables):
    """One loss + gradient evaluation with LIVE dropout (keep_prob 0.8, the DropoutWrapper around every GRU cell,
    rnn_class.py:146-152).  The masks the graph drew (its dropout/Floor nodes, per layer, direction and loop
    iteration) are recorded so that a trainer can replay them."""
    rng = np.random.RandomState(20260103)
    x = (rng.randn(DROP_BATCH, 35) * 1.5).astype(np.float32)
    y = (rng.rand(DROP_BATCH, 35) < 0.3).astype(np.float32)
    gi = G.GraphInterpreter(nodes, variables, np.float64, seed=11)
    floor = "recurrent_layer/stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/%s/while/dropout/Floor"
    for layer in range(3):
        for d in ("fw", "bw"):
            gi.trace[floor % (layer, d, d)] = []
    applies = sorted(k for k, n in nodes.items() if n.op == "ApplyRMSProp")
    names = [nodes[a].inputs[0][0] for a in applies]
    res = gi.run(["loss/Mean"] + ["%s:%d" % nodes[a].inputs[7] for a in applies],
                 {"data/Placeholder": x.reshape(-1, 35, 1), "data/Placeholder_1": y.reshape(-1, 35, 1),
                  "dropout": np.float32(DROP_KEEP)})
    out = {"drop_x": x, "drop_y": y, "drop_keep_prob": np.float64(DROP_KEEP), "drop_loss": np.float64(res[0])}
    masks = np.zeros((3, 2, DROP_BATCH, 35, 64), dtype=np.uint8)
    for layer in range(3):
        for di, d in enumerate(("fw", "bw")):
            seq = gi.trace[floor % (layer, d, d)]
            assert len(seq) == 35
            for it, m in enumerate(seq):                       # the bw loop walks the time-reversed sequence
                masks[layer, di, :, (34 - it) if d == "bw" else it, :] = m
    out["drop_masks"] = masks
    keep = ("conv1d/kernel", "conv1d_7/kernel", "batch_normalization_4/gamma", "final_fully_connected/kernel",
            "stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/gru_cell/gates/kernel",
            "stack_bidirectional_rnn/cell_1/bidirectional_rnn/bw/gru_cell/candidate/kernel",
            "stack_bidirectional_rnn/cell_2/bidirectional_rnn/bw/gru_cell/gates/bias")
    sums = {}
    for k, g in zip(names, res[1:]):
        g = np.asarray(g).reshape(np.asarray(variables[k]).shape)
        sums[k] = [float(g.sum()), float(np.abs(g).sum())]
        if k in keep:
            out["drop_grad/" + k] = g.astype(np.float32)
    out["drop_grad_sums_json"] = np.array(json.dumps(sums, sort_keys=True))
    return out


def summary(nodes):
    sub = G.inference_subgraph(nodes, FETCH)
    ops = {}
    for n in sub:
        ops[nodes[n].op] = ops.get(nodes[n].op, 0) + 1
    pick = lambda nd, keys: {k: nd.attr.get(k) for k in keys}     # noqa: E731
    convs = {n: pick(nodes[n], ("padding", "strides", "data_format", "dilations")) for n in sub if nodes[n].op == "Conv2D"}
    kernels = {n: [s for s, _ in nodes[n].inputs] for n in sub if nodes[n].op in ("Conv2D", "MatMul")}
    splits = {n: int(nodes[n].attr["num_split"]) for n in sub if nodes[n].op == "Split"}
    consts = {}
    for n in sub:
        nd = nodes[n]
        if nd.op == "Const" and (n.endswith("batchnorm/add/y") or n.endswith("split/split_dim") or n.endswith("concat/axis")
                                 and "gru_cell" in n):
            consts[n] = np.asarray(nd.attr["value"]).tolist()
    opt = {n: float(np.asarray(nodes[n].attr["value"])) for n in nodes
           if n.startswith("optimizer/RMSProp/") and nodes[n].op == "Const" and n.count("/") == 2}
    return {"meta_info": G.meta_info(PREFIX + ".meta"), "fetch": FETCH, "n_nodes_total": len(nodes), "n_nodes_inference": len(sub),
            "ops_inference": dict(sorted(ops.items())),
            "variables_inference": sorted(n for n in sub if nodes[n].op == "VariableV2"),
            "conv2d_attrs": convs, "matmul_conv_inputs": kernels, "split_num": splits, "constants": consts,
            "optimizer_constants": opt, "stage_nodes": STAGES}


def main():
    nodes = G.load_meta_graph(PREFIX + ".meta")
    variables = checkpoint.read_checkpoint(PREFIX)
    x, y = make_inputs()
    arrays = {"x": x, "y": y}
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        out, used = run(nodes, variables, x, y, dt)
        for k, v in out.items():
            if k in STAGES and k != "logits":
                if tag != "f64":
                    continue
                v = v[:STAGE_WINDOWS]                   # per-stage activations of the first windows only (file size)
            arrays["ckpt_%s_%s" % (tag, k)] = v
    rnd = dict(variables)
    rnd.update(O.random_weights(seed=RANDOM_SEED))
    out, _ = run(nodes, rnd, x, y, np.float64)
    arrays["rand_f64_probs"] = out["probs"]
    arrays["rand_f64_loss"] = out["loss"]
    arrays["random_seed"] = np.int64(RANDOM_SEED)
    np.savez_compressed(os.path.join(HERE, "graph_golden.npz"), **arrays)
    train = train_trajectory(nodes, variables)
    train.update(train_trajectory(nodes, variables, store32=True, tag="train32"))
    train.update(dropout_step(nodes, variables))
    np.savez_compressed(os.path.join(HERE, "graph_train_golden.npz"), **train)
    print("train losses", train["train_loss"])
    sm = summary(nodes)
    sm["ops_executed"] = dict(sorted(used.items()))
    with open(os.path.join(HERE, "graph_summary.json"), "w") as fh:
        json.dump(sm, fh, indent=1, sort_keys=True, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print("wrote graph_golden.npz (%d arrays), graph_summary.json; loss=%.6f acc=%.4f"
          % (len(arrays), float(arrays["ckpt_f64_loss"]), float(arrays["ckpt_f64_accuracy"])))


if __name__ == "__main__":
    main()
