"""The whole training step on the HIP kernels, without autograd (BASELINE config 5; reference ``RNN.train_network``,
catfish/models/rnn_class.py:201-210 = one ``optimizer.minimize(loss)``).

    conv stack forward            cf_res_train_forward
    biGRU layers forward          cf_gru_train_forward_dropout    (output dropout inside the kernels: no mask tensor)
    dense head + loss, fwd + bwd  cf_train_head                   (tf.losses.sigmoid_cross_entropy on the logits)
    biGRU layers backward         cf_gru_train_backward_dropout + cf_gru_train_wgrad
    conv stack backward           cf_res_train_backward
    optimizer + re-tiling         cf_opt_step                     (TF-1 RMSProp / Adam over ALL variables in one launch)

Every variable, its gradient and its two optimizer slots live in four flat device buffers of ONE layout, so the kernels
write their gradients straight into the optimizer's input and nothing is copied, concatenated or re-laid-out between
them:

    [ conv stack: per conv+BN unit  kernel | bias | gamma | beta | moving_mean | moving_variance ]
    [ biGRU layer 0: fw ( gates kernel | gates bias | candidate kernel | candidate bias ), bw ( ... ) ] [ layer 1 ] ...
    [ final_fully_connected kernel (128) | bias (1) ] [ 0.0 ]

The model's ``params`` dict (TF variable names) and the optimizer's slot dicts are VIEWS into these buffers, so checkpoint
I/O (``numpy_weights``, ``state_tf``) and the torch-autograd reference path keep working on the same storage.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .native_train import T, _stack_maps, frag_to_nat, nat_to_frag, res_unit_names

GRU_PRE = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell"


def _p(t):
    return C.c_void_p(t.data_ptr())


class NativeTrainStep(object):
    def __init__(self, net, opt, engine, keep_prob, seed=None):
        import torch
        self.seed = int(np.random.SeedSequence(seed).generate_state(1)[0])      # 32-bit dropout seed (fresh entropy when seed is None)
        self.torch = torch
        self.net, self.opt, self.engine = net, opt, engine
        self.lib, self.handle = engine._lib, engine._handle
        self.keep_prob = float(keep_prob)
        self.dev = net.device
        self.n_layers, self.n_blocks = net.n_layers, net.n_layers_res
        if self.n_blocks < 1:
            raise ValueError("the native training step needs the ResNetRNN type (n_layers_res >= 1)")
        self.kind = 1 if opt.choice == "Adam" else 0
        self._layout()
        self._rehome()
        self._bufs = None
        # weight gradients of layer L on a second stream, next to the backward recurrence of layer L - 1 (which needs only dX of layer
        # L): at the reference's batch of 256 windows the step is a chain of launches of 16 tiles on 256 CUs, and the twelve-wave wgrad
        # workgroups fit beside the four-wave recurrence ones.  A fork / join per layer in the captured graph.
        self.overlap_wgrad = True
        self._side = None

    # ------------------------------------------------------------------ flat layout
    def _layout(self):
        lib = self.lib
        self.entries = []                                   # (TF name, offset, shape)
        off = 0
        for unit in res_unit_names(self.n_blocks):
            for name in unit:
                shape = tuple(self.net.params[name].shape)
                self.entries.append((name, off, shape))
                off += int(np.prod(shape))
        self.n_res = off
        assert self.n_res == lib.cf_res_train_param_floats(self.n_blocks)
        self.cins = [32] + [128] * (self.n_layers - 1)
        self.gru_off = []
        remap = []                                          # virtual gather source (native_train._stack_maps) -> flat index
        self.zero_slot = None
        for layer, cin in enumerate(self.cins):
            rows = cin + 64
            self.gru_off.append(off)
            for d in ("fw", "bw"):
                pre = GRU_PRE % (layer, d)
                o_wg, o_bg, o_wc, o_bc = off, off + rows * 128, off + rows * 128 + 128, off + rows * 128 + 128 + rows * 64
                self.entries += [(pre + "/gates/kernel", o_wg, (rows, 128)), (pre + "/gates/bias", o_bg, (128,)),
                                 (pre + "/candidate/kernel", o_wc, (rows, 64)), (pre + "/candidate/bias", o_bc, (64,))]
                # the pack map's source block is [wg | wc | bg | bc | 0.0]
                remap += [np.arange(o_wg, o_wg + rows * 128), np.arange(o_wc, o_wc + rows * 64), np.arange(o_bg, o_bg + 128),
                          np.arange(o_bc, o_bc + 64), np.array([-1])]
                off = o_bc + 64
        self.head_off = off
        self.entries += [("final_fully_connected/kernel", off, (128, 1)), ("final_fully_connected/bias", off + 128, (1,))]
        off += 129
        self.zero_off = off
        self.n_total = off + 1
        remap = np.concatenate(remap)
        remap[remap < 0] = self.zero_off
        idx, scale, self.per_layer = _stack_maps(self.cins, self.dev)
        self.pack_idx = self.torch.from_numpy(remap[idx.cpu().numpy()].astype(np.int32)).to(self.dev)
        self.pack_scale = scale.contiguous()
        self.n_packed = int(self.pack_idx.numel())

    def _rehome(self):
        """Move the variables and the optimizer slots into the flat buffers; the dicts become views."""
        torch = self.torch
        net, opt = self.net, self.opt
        self.pflat = torch.zeros(self.n_total, dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.pflat)
        self.s1 = torch.ones_like(self.pflat) if self.kind == 0 else torch.zeros_like(self.pflat)   # TF: rms slot starts at 1
        self.s2 = torch.zeros_like(self.pflat)
        slots = (opt.ms, opt.mom) if self.kind == 0 else (opt.m, opt.v)
        with torch.no_grad():
            for name, off, shape in self.entries:
                n = int(np.prod(shape))
                old = net.params[name]
                view = self.pflat[off:off + n].view(shape)
                view.copy_(old.detach().reshape(shape))
                if old.requires_grad:
                    view.requires_grad_(True)
                    for flat, dic in zip((self.s1, self.s2), slots):
                        sv = flat[off:off + n].view(shape)
                        sv.copy_(dic[name].reshape(shape))
                        dic[name] = sv
                    opt.params[name] = view
                net.params[name] = view
        self.packed = torch.empty(self.n_packed, dtype=torch.float32, device=self.dev)
        self.retile()

    def retile(self):
        """packed biGRU weights (forward and backward tilings of every layer) from the flat variables."""
        # (a gather without an optimizer update, e.g. after loading variables: done on the torch side)
        self.packed.copy_(self.pflat[self.pack_idx.long()] * self.pack_scale)

    # ------------------------------------------------------------------ buffers per batch size
    def _alloc(self, n):
        torch = self.torch
        dev = self.dev
        lib = self.lib
        npad = (n + 15) // 16 * 16
        tiles = npad // 16
        b = {"n": n, "npad": npad, "tiles": tiles}
        f32 = dict(dtype=torch.float32, device=dev)
        b["x"] = torch.zeros(n, T, **f32)
        b["y"] = torch.zeros(npad, T, **f32)
        b["z"] = torch.empty(4 * self.n_blocks, n * T, 32, **f32)
        b["res_out"] = torch.zeros(npad, T, 32, **f32)
        b["y_frag"] = [torch.empty(tiles, T, 8, 64, 4, **f32) for _ in range(self.n_layers)]
        b["stash"] = [torch.empty(tiles, T, 2, 12, 64, 4, **f32) for _ in range(self.n_layers)]
        b["y_drop"] = [torch.empty(tiles, T, 8, 64, 4, **f32) for _ in range(self.n_layers)] if self.keep_prob < 1.0 else None
        b["dy_head"] = torch.empty(tiles, T, 8, 64, 4, **f32)
        b["dx"] = [torch.empty(2, tiles, T, c // 16, 64, 4, **f32) for c in self.cins]
        # per layer: layer L's pre-activation gradients and partial sums are still being read (wgrad, side stream) while layer L - 1's are written
        b["da"] = [torch.empty(tiles, T, 2, 12, 64, 4, **f32) for _ in self.cins]
        b["head_ws"] = torch.empty(int(lib.cf_train_head_workspace_floats(self.handle, npad)), **f32)
        b["wgrad_ws"] = [torch.empty(int(lib.cf_gru_wgrad_workspace_floats(self.handle, c, npad)), **f32) for c in self.cins]
        b["res_ws"] = torch.empty(int(lib.cf_res_train_workspace_floats(self.n_blocks, n)), **f32)
        b["loss"] = torch.zeros(1, **f32)
        return b

    # ------------------------------------------------------------------ one step (all launches on the current stream)
    def run(self, b, keep_prob=None, masks=None, update=True):
        torch = self.torch
        lib, h = self.lib, self.handle
        kp = self.keep_prob if keep_prob is None else float(keep_prob)
        n, npad = b["n"], b["npad"]
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        pf, gf = self.pflat, self.gflat
        # ---- forward
        N.check(lib.cf_res_train_forward(h, self.n_blocks, _p(pf), _p(b["x"]), _p(b["z"]), _p(b["res_out"]), n, stream))
        cur = nat_to_frag(b["res_out"])
        # output dropout: inside the kernels (hash of seed, layer, optimizer step, element) unless explicit masks are replayed
        drop = None
        in_kernel = kp < 1.0 and masks is None
        if kp < 1.0 and masks is not None:
            from .native_train import dropout_scale_frag
            drop = [dropout_scale_frag(npad, kp, self.dev, masks, layer) for layer in range(self.n_layers)]
        if in_kernel and b.get("y_drop") is None:
            b["y_drop"] = [torch.empty_like(b["y_frag"][0]) for _ in range(self.n_layers)]
        step_ptr = _p(self.opt.t)
        inputs = []
        for layer in range(self.n_layers):
            fo, fn, _, _ = self.per_layer[layer]
            if in_kernel:
                N.check(lib.cf_gru_train_forward_dropout(h, self.cins[layer], _p(self.packed[fo:fo + fn]), _p(cur), _p(b["y_frag"][layer]),
                                                         _p(b["stash"][layer]), npad, _p(b["y_drop"][layer]), kp, self.seed, layer,
                                                         step_ptr, stream))
            else:
                N.check(lib.cf_gru_train_forward(h, self.cins[layer], _p(self.packed[fo:fo + fn]), _p(cur), _p(b["y_frag"][layer]),
                                                 _p(b["stash"][layer]), npad, stream))
            inputs.append(cur)
            if in_kernel:
                cur = b["y_drop"][layer]
            else:
                cur = b["y_frag"][layer] if drop is None else b["y_frag"][layer] * drop[layer]
        # ---- dense head + loss, forward and backward
        ho = self.head_off
        N.check(lib.cf_train_head(h, _p(cur), _p(pf[ho:ho + 128]), _p(pf[ho + 128:ho + 129]), _p(b["y"]), n, _p(b["dy_head"]), None,
                                  _p(b["head_ws"]), int(b["head_ws"].numel()), _p(gf[ho:ho + 129]), _p(b["loss"]), stream))
        # ---- backward through the biGRU layers
        main = torch.cuda.current_stream(self.dev)
        side = None
        if self.overlap_wgrad:
            if self._side is None:
                self._side = torch.cuda.Stream(self.dev)
            side = self._side
        g0, g1 = b["dy_head"], None
        for layer in range(self.n_layers - 1, -1, -1):
            cin = self.cins[layer]
            _, _, bo, bn = self.per_layer[layer]
            dx = b["dx"][layer]
            N.check(lib.cf_gru_train_backward_dropout(h, cin, _p(self.packed[bo:bo + bn]), _p(b["y_frag"][layer]), _p(b["stash"][layer]),
                                                      _p(g0), None if g1 is None else _p(g1), None if drop is None else _p(drop[layer]),
                                                      _p(dx), _p(b["da"][layer]), npad, kp if in_kernel else 1.0, self.seed, layer, step_ptr,
                                                      stream))
            go = self.gru_off[layer]
            wstream = stream
            if side is not None:
                side.wait_stream(main)                      # fork: everything up to this layer's backward recurrence
                wstream = C.c_void_p(side.cuda_stream)
            N.check(lib.cf_gru_train_wgrad(h, cin, _p(inputs[layer]), _p(b["y_frag"][layer]), _p(b["stash"][layer]), _p(b["da"][layer]), npad,
                                           _p(b["wgrad_ws"][layer]), int(b["wgrad_ws"][layer].numel()), _p(gf[go:]), wstream))
            g0, g1 = dx[0], dx[1]
        d_out = frag_to_nat(g0 + g1)[:n].contiguous()
        N.check(lib.cf_res_train_backward(h, self.n_blocks, _p(pf), _p(b["x"]), _p(b["z"]), _p(d_out), _p(b["res_ws"]),
                                          int(b["res_ws"].numel()), _p(gf), n, stream))
        if side is not None:
            main.wait_stream(side)                          # join: the optimizer reads every gradient
        # ---- optimizer over every variable + re-tiling of the biGRU weights
        if update:
            N.check(lib.cf_opt_step(h, self.kind, _p(pf), _p(gf), _p(self.s1), _p(self.s2), self.n_total - 1, float(self.opt.lr),
                                    _p(self.opt.t), _p(self.pack_idx), _p(self.pack_scale), _p(self.packed), self.n_packed, stream))
        return b["loss"]

    def load_batch(self, b, x, y):
        """The batch into the step's static device buffers.  (Pinned staging was measured in round 6 and dropped: 0.764-0.781 ms per
        256-window step against 0.754-0.762 with these plain copies, `profiles/r06_train_pinned_vs_pageable.log` -- the extra host copy
        and event cost more than the synchronous 36 KB copies.)"""
        torch = self.torch
        n = b["n"]
        b["x"].copy_(torch.as_tensor(np.asarray(x), dtype=torch.float32).reshape(n, T), non_blocking=True)
        b["y"][:n].copy_(torch.as_tensor(np.asarray(y), dtype=torch.float32).reshape(n, T), non_blocking=True)

    def dropout_scales(self, n, keep_prob=None):
        """The mask / keep_prob tensors the kernels apply at the CURRENT optimizer step, as {(layer, "fw"|"bw"): 0/1 array
        [n, 35, 64]} (tests replay them through the torch reference path)."""
        torch = self.torch
        kp = self.keep_prob if keep_prob is None else float(keep_prob)
        npad = (n + 15) // 16 * 16
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        out = {}
        for layer in range(self.n_layers):
            sc = torch.empty(npad // 16, T, 8, 64, 4, dtype=torch.float32, device=self.dev)
            N.check(self.lib.cf_dropout_scale(self.handle, kp, self.seed, layer, _p(self.opt.t), npad, _p(sc), stream))
            m = (frag_to_nat(sc)[:n] > 0).to(torch.float32).cpu().numpy()
            out[(layer, "fw")], out[(layer, "bw")] = m[:, :, :64], m[:, :, 64:]
        return out

    def grads(self):
        """{TF name: gradient view} of the last step (trainable variables only)."""
        return {name: self.gflat[off:off + int(np.prod(shape))].view(shape) for name, off, shape in self.entries
                if self.net.params[name].requires_grad}
