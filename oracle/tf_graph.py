"""Numpy interpreter for the reference's OWN TensorFlow graph (test infrastructure only).

The reference ships the MetaGraphDef its TF-1.10 session executed
(``catfish/ResNetRNN/checkpoints/ckpnt-30000.meta``, written by ``tf.train.Saver`` at
rnn_class.py:48 / train.py).  This module decodes that protobuf with a wire-format parser
(nothing from the file is executed) and evaluates the inference subgraph
``data/Placeholder -> accuracy/Sigmoid`` (rnn_class.py:84,213-216) node by node with numpy
kernels, including the ``dynamic_rnn`` while-loops (Enter / Merge / Switch / NextIteration / Exit
and the TensorArray ops).

What this pins and what it does not:

* pinned by the reference's file: the op sequence, every attribute (padding, strides,
  data_format, split axis, transpose permutations, the dropout wiring, epsilon constants),
  which variable feeds which op, gate order and the exact GRU-cell dataflow, time reversal of the
  backward direction, the final reshape;
* still a restatement: the arithmetic of each individual TF op kernel (Conv2D, MatMul, Sigmoid,
  ...), written here from TF-1.10's published op definitions.  TensorFlow itself is not
  importable in this image.

Only ``tests/`` and ``tests/golden/make_graph_golden.py`` use it; the product never does.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np


# ------------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def parse_message(buf) -> Dict[int, list]:
    """field number -> list of raw values (ints for varint/fixed, bytes for length-delimited)."""
    fields: Dict[int, list] = {}
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        fields.setdefault(fno, []).append(val)
    return fields


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _repeated_varints(vals):
    """A repeated integer field may arrive packed (bytes) or one varint per entry."""
    out = []
    for v in vals:
        if isinstance(v, bytes):
            pos = 0
            while pos < len(v):
                x, pos = _varint(v, pos)
                out.append(_signed(x))
        else:
            out.append(_signed(v))
    return out


def _repeated_fixed(vals, fmt, size):
    out = []
    for v in vals:
        if isinstance(v, bytes):
            out.extend(struct.unpack("<%d%s" % (len(v) // size, fmt), v))
        else:
            out.append(struct.unpack("<" + fmt, struct.pack("<I" if size == 4 else "<Q", v))[0])
    return out


# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 7, 9, 10
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64, DT_BOOL: np.bool_}


def _shape(buf):
    msg = parse_message(buf)
    if msg.get(3, [0])[0]:
        return None                                        # unknown rank
    return tuple(_signed(parse_message(d).get(1, [0])[0]) for d in msg.get(2, []))


def _tensor(buf):
    """TensorProto (tensorflow/core/framework/tensor.proto) -> ndarray."""
    msg = parse_message(buf)
    dtype = msg[1][0]
    shape = _shape(msg[2][0]) if 2 in msg else ()
    if dtype == DT_STRING:
        return np.array([s for s in msg.get(8, [])], dtype=object).reshape(shape if shape else ())
    npdt = _NP[dtype]
    count = int(np.prod(shape)) if shape else 1
    if 4 in msg and len(msg[4][0]):
        arr = np.frombuffer(msg[4][0], dtype=npdt).copy()
    else:
        if dtype == DT_FLOAT:
            vals = _repeated_fixed(msg.get(5, []), "f", 4)
        elif dtype == DT_DOUBLE:
            vals = _repeated_fixed(msg.get(6, []), "d", 8)
        elif dtype == DT_INT32:
            vals = _repeated_varints(msg.get(7, []))
        elif dtype == DT_INT64:
            vals = _repeated_varints(msg.get(10, []))
        else:
            vals = [bool(v) for v in _repeated_varints(msg.get(11, []))]
        if not vals:
            vals = [0]
        arr = np.array(vals, dtype=npdt)
        if arr.size == 1 and count != 1:                   # splat encoding
            arr = np.full(count, arr[0], dtype=npdt)
        elif arr.size < count:                             # last value repeats
            arr = np.concatenate([arr, np.full(count - arr.size, arr[-1], dtype=npdt)])
    return arr.reshape(shape)


def _attr(buf):
    """AttrValue (attr_value.proto) -> python value."""
    msg = parse_message(buf)
    if 2 in msg:
        return msg[2][0].decode("utf-8", "replace")
    if 3 in msg:
        return _signed(msg[3][0])
    if 4 in msg:
        return struct.unpack("<f", struct.pack("<I", msg[4][0]))[0]
    if 5 in msg:
        return bool(msg[5][0])
    if 6 in msg:
        return ("dtype", msg[6][0])
    if 7 in msg:
        return ("shape", _shape(msg[7][0]))
    if 8 in msg:
        return _tensor(msg[8][0])
    if 1 in msg:
        lst = parse_message(msg[1][0])
        if 3 in lst:
            return _repeated_varints(lst[3])
        if 2 in lst:
            return [s.decode("utf-8", "replace") for s in lst[2]]
        if 4 in lst:
            return _repeated_fixed(lst[4], "f", 4)
        if 6 in lst:
            return [("dtype", v) for v in _repeated_varints(lst[6])]
        if 7 in lst:
            return [("shape", _shape(s)) for s in lst[7]]
        return []
    return None


class Node(object):
    __slots__ = ("name", "op", "inputs", "controls", "attr")

    def __init__(self, name, op, inputs, controls, attr):
        self.name, self.op, self.inputs, self.controls, self.attr = name, op, inputs, controls, attr


def load_meta_graph(path: str) -> Dict[str, Node]:
    """MetaGraphDef file -> {node name: Node}.  MetaGraphDef.graph_def = field 2, GraphDef.node = field 1."""
    with open(path, "rb") as fh:
        meta = parse_message(fh.read())
    graph = parse_message(meta[2][0])
    nodes: Dict[str, Node] = {}
    for raw in graph[1]:
        nd = parse_message(raw)
        name = nd[1][0].decode()
        op = nd[2][0].decode()
        inputs: List[Tuple[str, int]] = []
        controls: List[str] = []
        for i in nd.get(3, []):
            s = i.decode()
            if s.startswith("^"):
                controls.append(s[1:])
            elif ":" in s:
                a, b = s.rsplit(":", 1)
                inputs.append((a, int(b)))
            else:
                inputs.append((s, 0))
        attr = {}
        for entry in nd.get(5, []):
            kv = parse_message(entry)
            attr[kv[1][0].decode()] = _attr(kv[2][0]) if 2 in kv else None
        nodes[name] = Node(name, op, inputs, controls, attr)
    return nodes


def meta_info(path: str) -> dict:
    """tensorflow_version etc. from MetaGraphDef.meta_info_def (field 1)."""
    with open(path, "rb") as fh:
        meta = parse_message(fh.read())
    info = parse_message(meta[1][0])
    return {"tensorflow_version": info.get(5, [b""])[0].decode(), "tensorflow_git_version": info.get(6, [b""])[0].decode()}


# ------------------------------------------------------------------------------- interpreter
class _TensorArray(object):
    def __init__(self, size):
        self.items = [None] * int(size)


class _Ctx(object):
    def __init__(self, outer=None, merges=None):
        self.memo = {}
        self.outer = outer
        self.merges = merges or {}


class GraphInterpreter(object):
    """Evaluates tensors of a TF-1.x GraphDef with numpy.

    ``variables``: {variable name: ndarray} (the checkpoint).  ``float_dtype`` = np.float32 runs
    the graph as TF does; np.float64 promotes every float constant / variable / feed so the
    result is the graph's exact-arithmetic value (used to compare against the fp64 oracle).
    """

    def __init__(self, nodes: Dict[str, Node], variables: Dict[str, np.ndarray], float_dtype=np.float32, seed=0):
        self.nodes = nodes
        self.variables = variables
        self.fdt = float_dtype
        self.rng = np.random.RandomState(seed)
        self.ops_used = {}
        # while-loop frames
        self.frame_enters = {}
        for n in nodes.values():
            if n.op == "Enter":
                self.frame_enters.setdefault(n.attr["frame_name"], []).append(n.name)
        self.merge_frame = {}
        self.frame_merges = {}
        for n in nodes.values():
            if n.op == "Merge":
                for src, _ in n.inputs:
                    if nodes[src].op == "Enter":
                        fr = nodes[src].attr["frame_name"]
                        self.merge_frame[n.name] = fr
                        self.frame_merges.setdefault(fr, []).append(n.name)
        self.frame_cond = {}
        self.exit_merge = {}
        self.frame_exits = {}
        for n in nodes.values():
            if n.op == "Switch":
                m = self._through_identity(n.inputs[0][0])
                if m in self.merge_frame:
                    self.frame_cond[self.merge_frame[m]] = n.inputs[1]
            elif n.op == "Exit":
                sw = self._through_identity(n.inputs[0][0])
                if nodes[sw].op == "Switch":
                    m = self._through_identity(nodes[sw].inputs[0][0])
                    if m in self.merge_frame:
                        self.exit_merge[n.name] = m
                        self.frame_exits.setdefault(self.merge_frame[m], []).append(n.name)
        # gradient stacks: the pushes sit in the forward loop and nothing in that loop consumes them
        self.frame_pushes = {}
        for n in nodes.values():
            if n.op == "StackPushV2" and nodes[n.inputs[0][0]].op == "Enter":
                self.frame_pushes.setdefault(nodes[n.inputs[0][0]].attr["frame_name"], []).append(n.name)
        self.grad_arrays = {}
        self.updates = {}
        self.trace = {}            # {node name: []}: every value the node takes (once per loop iteration), in run order

    def _through_identity(self, name):
        while self.nodes[name].op == "Identity":
            name = self.nodes[name].inputs[0][0]
        return name

    def _f(self, arr):
        arr = np.asarray(arr)
        return arr.astype(self.fdt) if arr.dtype.kind == "f" else arr

    def run(self, fetch, feeds: Dict[str, np.ndarray]):
        """fetch: a tensor name ("node" or "node:k") or a list of them (evaluated in ONE pass, like one
        session.run: loops, stacks and random ops execute once)."""
        import sys
        if sys.getrecursionlimit() < 50000:
            sys.setrecursionlimit(50000)
        ctx = _Ctx()
        for k, v in feeds.items():
            ctx.memo[(k, 0)] = self._f(v)
        self.grad_arrays = {}
        self.updates = {}
        many = isinstance(fetch, (list, tuple))
        out = []
        for f in (fetch if many else [fetch]):
            name, idx = f.rsplit(":", 1) if ":" in f else (f, "0")
            out.append(self.eval((name, int(idx)), ctx))
        return out if many else out[0]

    # -- evaluation ------------------------------------------------------------------------
    def eval(self, tensor, ctx):
        c = ctx
        while c is not None:
            if tensor in c.memo:
                return c.memo[tensor]
            c = c.outer
        name, idx = tensor
        node = self.nodes[name]
        op = node.op
        self.ops_used[op] = self.ops_used.get(op, 0) + 1
        if op == "Enter":
            root = ctx
            while root.outer is not None:
                root = root.outer
            val = self.eval(node.inputs[0], root)
            root.memo[tensor] = val
            return val
        if op == "Merge":
            if name in ctx.merges:
                return ctx.merges[name]
            raise RuntimeError("Merge %s evaluated outside its loop" % name)
        if op == "Exit":
            self._run_loop(self.merge_frame[self.exit_merge[name]], ctx)
            return ctx.memo[tensor]
        if op == "NextIteration":
            raise RuntimeError("NextIteration reached directly: %s" % name)
        args = [self.eval(t, ctx) for t in node.inputs]
        outs = self._kernel(node, args)
        for i, o in enumerate(outs):
            ctx.memo[(name, i)] = o
        if name in self.trace:
            self.trace[name].append(outs[0])
        return outs[idx]

    def _run_loop(self, frame, ctx):
        merges = self.frame_merges[frame]
        values = {}
        nexts = {}
        for m in merges:
            for src, sidx in self.nodes[m].inputs:
                if self.nodes[src].op == "Enter":
                    values[m] = self.eval((src, sidx), ctx)
                else:
                    nexts[m] = self.nodes[src].inputs[0]
        guard = 0
        while True:
            it = _Ctx(outer=ctx, merges=values)
            if not bool(self.eval(self.frame_cond[frame], it)):
                break
            for push in self.frame_pushes.get(frame, ()):
                self.eval((push, 0), it)
            values = {m: self.eval(nexts[m], it) for m in merges}
            guard += 1
            if guard > 100000:
                raise RuntimeError("while loop %s does not terminate" % frame)
        for ex in self.frame_exits.get(frame, ()):
            ctx.memo[(ex, 0)] = values[self.exit_merge[ex]]

    # -- op kernels (TF-1.10 op definitions) -------------------------------------------------
    def _kernel(self, node, a):
        op, at = node.op, node.attr
        if op == "Placeholder":
            raise KeyError("placeholder %s was not fed" % node.name)
        if op == "Const":
            return [self._f(at["value"])]
        if op == "VariableV2":
            return [self._f(self.variables[node.name])]
        if op in ("Identity", "LoopCond", "StopGradient"):
            return [a[0]]
        if op == "Switch":
            return [a[0], a[0]]
        if op == "ExpandDims":
            return [np.expand_dims(a[0], int(a[1]))]
        if op == "Squeeze":
            dims = at.get("squeeze_dims") or None
            return [np.squeeze(a[0], axis=tuple(dims) if dims else None)]
        if op == "Conv2D":
            return [self._conv2d(a[0], a[1], at)]
        if op == "BiasAdd":
            assert at.get("data_format", "NHWC") == "NHWC"
            return [a[0] + a[1]]
        if op == "Add":
            return [a[0] + a[1]]
        if op == "Sub":
            return [a[0] - a[1]]
        if op == "Mul":
            return [a[0] * a[1]]
        if op == "RealDiv":
            return [a[0] / a[1]]
        if op == "Maximum":
            return [np.maximum(a[0], a[1])]
        if op == "Minimum":
            return [np.minimum(a[0], a[1])]
        if op == "Rsqrt":
            return [(1.0 / np.sqrt(a[0])).astype(a[0].dtype)]
        if op == "Relu":
            return [np.maximum(a[0], 0).astype(a[0].dtype)]
        if op == "Sigmoid":
            with np.errstate(over="ignore"):
                return [(1.0 / (1.0 + np.exp(-a[0]))).astype(a[0].dtype)]
        if op == "Tanh":
            return [np.tanh(a[0])]
        if op == "Floor":
            return [np.floor(a[0])]
        if op == "Shape":
            return [np.array(np.shape(a[0]), dtype=np.int32)]
        if op == "Reshape":
            return [np.reshape(a[0], [int(v) for v in a[1]])]
        if op == "Transpose":
            return [np.transpose(a[0], [int(v) for v in a[1]])]
        if op == "ReverseV2":
            return [np.flip(a[0], axis=tuple(int(v) for v in np.atleast_1d(a[1])))]
        if op == "ConcatV2":
            return [np.concatenate(a[:-1], axis=int(a[-1]))]
        if op == "Split":
            return list(np.split(a[1], int(at["num_split"]), axis=int(a[0])))
        if op == "MatMul":
            x = a[0].T if at.get("transpose_a") else a[0]
            y = a[1].T if at.get("transpose_b") else a[1]
            return [x @ y]
        if op == "Fill":
            return [np.full([int(v) for v in a[0]], a[1], dtype=np.asarray(a[1]).dtype)]
        if op == "Range":
            return [np.arange(a[0], a[1], a[2], dtype=np.asarray(a[0]).dtype)]
        if op == "Less":
            return [np.less(a[0], a[1])]
        if op == "GreaterEqual":
            return [np.greater_equal(a[0], a[1])]
        if op == "LogicalAnd":
            return [np.logical_and(a[0], a[1])]
        if op == "Select":
            return [np.where(a[0], a[1], a[2])]
        if op == "ZerosLike":
            return [np.zeros_like(a[0])]
        if op == "Neg":
            return [-a[0]]
        if op == "Exp":
            return [np.exp(a[0])]
        if op == "Log1p":
            return [np.log1p(a[0])]
        if op == "Round":                                   # TF rounds half to even, like numpy
            return [np.round(a[0])]
        if op == "Equal":
            return [np.equal(a[0], a[1])]
        if op == "Greater":
            return [np.greater(a[0], a[1])]
        if op == "Cast":
            dst = at["DstT"][1]
            return [np.asarray(a[0]).astype(self.fdt if dst in (DT_FLOAT, DT_DOUBLE) else _NP[dst])]
        if op in ("Sum", "Mean"):
            axes = tuple(int(v) for v in np.atleast_1d(a[1]))
            fn = np.sum if op == "Sum" else np.mean
            if not axes and np.ndim(a[1]) == 1:             # empty reduction_indices: nothing is reduced
                return [np.asarray(a[0])]
            return [fn(a[0], axis=axes, keepdims=bool(at.get("keep_dims")), dtype=np.asarray(a[0]).dtype)]
        if op in ("NoOp", "ControlTrigger"):
            return [None]
        # ---- gradient graph -------------------------------------------------------------------
        if op == "StackV2":
            return [[]]
        if op == "StackPushV2":
            a[0].append(a[1])
            return [a[1]]
        if op == "StackPopV2":
            return [a[0].pop()]
        if op == "TensorArrayGradV3":                      # one gradient array per (forward array, source)
            key = (id(a[0]), at["source"])
            if key not in self.grad_arrays:
                self.grad_arrays[key] = _TensorArray(len(a[0].items))
                self.grad_arrays[key].is_grad = True
            return [self.grad_arrays[key], np.float32(0)]
        if op == "AddN":
            out = a[0]
            for v in a[1:]:
                out = out + v
            return [out]
        if op == "Reciprocal":
            return [1.0 / a[0]]
        if op == "FloorMod":
            return [np.mod(a[0], a[1])]
        if op == "InvertPermutation":
            return [np.argsort(np.asarray(a[0])).astype(np.asarray(a[0]).dtype)]
        if op == "ShapeN":
            return [np.array(np.shape(v), dtype=np.int32) for v in a]
        if op == "ConcatOffset":                            # offsets of each input along concat_dim
            dim = int(a[0])
            outs, off = [], 0
            for shp in a[1:]:
                o = np.zeros(len(shp), dtype=np.int32)
                o[dim] = off
                off += int(shp[dim])
                outs.append(o)
            return outs
        if op == "Slice":
            idx = tuple(slice(int(b), None if int(sz) == -1 else int(b) + int(sz)) for b, sz in zip(a[1], a[2]))
            return [np.asarray(a[0])[idx]]
        if op == "Tile":
            return [np.tile(a[0], [int(v) for v in a[1]])]
        if op == "BroadcastGradientArgs":
            return list(self._broadcast_gradient_args(a[0], a[1]))
        if op == "BiasAddGrad":
            g = np.asarray(a[0])
            return [g.reshape(-1, g.shape[-1]).sum(axis=0)]
        if op == "ReluGrad":
            return [a[0] * (a[1] > 0)]
        if op == "TanhGrad":                                # (y, dy)
            return [a[1] * (1.0 - a[0] * a[0])]
        if op == "SigmoidGrad":                             # (y, dy)
            return [a[1] * a[0] * (1.0 - a[0])]
        if op == "Conv2DBackpropInput":
            return [self._conv2d_backprop_input([int(v) for v in a[0]], a[1], a[2], at)]
        if op == "Conv2DBackpropFilter":
            return [self._conv2d_backprop_filter(a[0], [int(v) for v in a[1]], a[2], at)]
        if op == "ApplyRMSProp":                            # training_ops.cc: ms, mom, var in that order
            var, ms, mom, lr, rho, momentum, eps, g = a
            ms_new = ms + (g * g - ms) * (1.0 - rho)
            mom_new = mom * momentum + lr * g / np.sqrt(ms_new + eps)
            var_new = var - mom_new
            for src, val in zip(node.inputs[:3], (var_new, ms_new, mom_new)):
                self.updates[src[0]] = val
            return [var_new]
        if op == "Max":
            return [np.max(a[0], axis=tuple(int(v) for v in np.atleast_1d(a[1])), keepdims=bool(at.get("keep_dims")))]
        if op == "StridedSlice":
            return [self._strided_slice(a[0], a[1], a[2], a[3], at)]
        if op == "RandomUniform":
            return [self.rng.random_sample([int(v) for v in a[0]]).astype(self.fdt)]
        if op == "TensorArrayV3":
            return [_TensorArray(a[0]), np.float32(0)]
        if op == "TensorArrayScatterV3":
            ta, idx, val = a[0], a[1], a[2]
            for k, i in enumerate(np.asarray(idx).tolist()):
                ta.items[i] = val[k]
            return [np.float32(0)]
        if op == "TensorArrayReadV3":
            return [a[0].items[int(a[1])]]
        if op == "TensorArrayWriteV3":
            i = int(a[1])
            if getattr(a[0], "is_grad", False) and a[0].items[i] is not None:
                a[0].items[i] = a[0].items[i] + a[2]        # gradient arrays aggregate repeated writes
            else:
                a[0].items[i] = a[2]
            return [np.float32(0)]
        if op == "TensorArrayGatherV3":
            return [np.stack([a[0].items[i] for i in np.asarray(a[1]).tolist()])]
        if op == "TensorArraySizeV3":
            return [np.int32(len(a[0].items))]
        raise NotImplementedError("op %s (%s)" % (op, node.name))

    @staticmethod
    def _conv2d(x, w, at):
        """Conv2D, NHWC, unit strides / dilations: out[n,h,w,:] = sum_{i,j} x[n,h+i-pt,w+j-pl,:] @ w[i,j]."""
        assert at.get("data_format", "NHWC") == "NHWC", at
        assert list(at.get("strides", [1, 1, 1, 1])) == [1, 1, 1, 1], at
        assert list(at.get("dilations", [1, 1, 1, 1]) or [1, 1, 1, 1]) == [1, 1, 1, 1], at
        kh, kw = w.shape[0], w.shape[1]
        n, h, wd, _ = x.shape
        if at["padding"] == "SAME":                     # TF: total = k - 1, the smaller half in front
            pt, pl = (kh - 1) // 2, (kw - 1) // 2
            xp = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
            oh, ow = h, wd
        elif at["padding"] == "VALID":
            xp = x
            oh, ow = h - kh + 1, wd - kw + 1
        else:
            raise NotImplementedError(at["padding"])
        out = np.zeros((n, oh, ow, w.shape[3]), dtype=np.result_type(x, w))
        for i in range(kh):
            for j in range(kw):
                out += xp[:, i:i + oh, j:j + ow, :] @ w[i, j]
        return out

    @staticmethod
    def _same_geometry(x_shape, w_shape, at):
        assert at.get("data_format", "NHWC") == "NHWC" and list(at.get("strides", [1, 1, 1, 1])) == [1, 1, 1, 1], at
        assert at["padding"] == "SAME", at
        kh, kw = w_shape[0], w_shape[1]
        return kh, kw, (kh - 1) // 2, (kw - 1) // 2

    @classmethod
    def _conv2d_backprop_input(cls, x_shape, w, dy, at):
        """dx[n,h,w,:] = sum_{i,j} dy[n,h-i+pt,w-j+pl,:] @ w[i,j]^T  (transpose of _conv2d)."""
        kh, kw, pt, pl = cls._same_geometry(x_shape, w.shape, at)
        n, h, wd, _ = x_shape
        dxp = np.zeros((n, h + kh - 1, wd + kw - 1, w.shape[2]), dtype=np.result_type(dy, w))
        for i in range(kh):
            for j in range(kw):
                dxp[:, i:i + h, j:j + wd, :] += dy @ w[i, j].T
        return dxp[:, pt:pt + h, pl:pl + wd, :]

    @classmethod
    def _conv2d_backprop_filter(cls, x, w_shape, dy, at):
        """dw[i,j] = sum_{n,h,w} x[n,h+i-pt,w+j-pl,:]^T dy[n,h,w,:]."""
        kh, kw, pt, pl = cls._same_geometry(x.shape, w_shape, at)
        n, h, wd, cin = x.shape
        xp = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
        dw = np.zeros(w_shape, dtype=np.result_type(x, dy))
        d2 = dy.reshape(-1, dy.shape[-1])
        for i in range(kh):
            for j in range(kw):
                dw[i, j] = xp[:, i:i + h, j:j + wd, :].reshape(-1, cin).T @ d2
        return dw

    @staticmethod
    def _broadcast_gradient_args(s0, s1):
        """Reduction axes that undo numpy-style broadcasting of shapes s0 and s1 (ops/array_ops.cc)."""
        s0, s1 = [int(v) for v in s0], [int(v) for v in s1]
        rank = max(len(s0), len(s1))
        p0, p1 = [1] * (rank - len(s0)) + s0, [1] * (rank - len(s1)) + s1
        r0 = [d for d in range(rank) if p0[d] == 1 and (p1[d] != 1 or d < rank - len(s0))]
        r1 = [d for d in range(rank) if p1[d] == 1 and (p0[d] != 1 or d < rank - len(s1))]
        return np.array(r0, dtype=np.int32), np.array(r1, dtype=np.int32)

    @staticmethod
    def _strided_slice(x, begin, end, strides, at):
        if at.get("ellipsis_mask") or at.get("new_axis_mask"):
            raise NotImplementedError("StridedSlice ellipsis/new_axis")
        bm, em, sm = int(at.get("begin_mask") or 0), int(at.get("end_mask") or 0), int(at.get("shrink_axis_mask") or 0)
        index = []
        for d in range(len(begin)):
            b, e, s = int(begin[d]), int(end[d]), int(strides[d])
            if sm & (1 << d):
                index.append(b)
            else:
                index.append(slice(None if bm & (1 << d) else b, None if em & (1 << d) else e, s))
        return np.asarray(x)[tuple(index)]


def inference_subgraph(nodes: Dict[str, Node], fetch: str) -> List[str]:
    """Names of all nodes the fetch depends on (data and control edges, through NextIteration)."""
    seen, stack = set(), [fetch]
    while stack:
        x = stack.pop()
        if x in seen:
            continue
        seen.add(x)
        nd = nodes[x]
        stack.extend(s for s, _ in nd.inputs)
        stack.extend(nd.controls)
    return sorted(seen)
