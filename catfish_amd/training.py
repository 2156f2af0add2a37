"""Training step (BASELINE config 5) -- the reference's ``train_network`` on PyTorch-ROCm autograd.

Reference: ``RNN.train_network`` (catfish/models/rnn_class.py:201-210) runs one
``optimizer.minimize(loss)`` step; loss = mean sigmoid cross-entropy over all B*35 outputs
(:74-79); optimizer = Adam or RMSProp with TF-1 defaults (:62-71); dropout with
``keep_prob`` on the GRU OUTPUTS only (:151-154); batch-norm stays in inference mode while
training (``training`` is never passed, resnet_class.py:61) so only gamma/beta learn and the
moving statistics never move.

The graph is restated with torch ops in the TF variable layout (same names/shapes as the
checkpoint), so trained weights go straight back into the HIP engine or a checkpoint-V2 bundle
(``checkpoint.write_checkpoint``).  The optimizer updates follow TF's formulas, not torch.optim's
(epsilon placement and slot initialisation differ).
"""
from __future__ import annotations

import numpy as np


def _names(n_layers, n_layers_res):
    conv = lambda j: "conv1d" if j == 0 else "conv1d_%d" % j            # noqa: E731
    bn = lambda j: "batch_normalization" if j == 0 else "batch_normalization_%d" % j   # noqa: E731
    return conv, bn


class TorchResNetRNN(object):
    """Differentiable restatement of the ResNetRNN graph; parameters are a dict of torch tensors."""

    def __init__(self, weights, n_layers, n_layers_res, device="cpu", dtype=None):
        import torch
        self.torch = torch
        self.dtype = dtype or torch.float32
        self.device = torch.device(device)
        self.n_layers = int(n_layers)
        self.n_layers_res = int(n_layers_res)
        self.params = {}
        for k, v in weights.items():
            t = torch.tensor(np.asarray(v), dtype=self.dtype, device=self.device)
            t.requires_grad_(not k.endswith(("moving_mean", "moving_variance")))
            self.params[k] = t

    def trainable(self):
        return {k: v for k, v in self.params.items() if v.requires_grad}

    def numpy_weights(self):
        return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in self.params.items()}

    # ---- forward ---------------------------------------------------------------------------
    def _conv_bn(self, x, j):
        torch = self.torch
        F = torch.nn.functional
        conv, bn = _names(self.n_layers, self.n_layers_res)
        p = self.params
        k = p[conv(j) + "/kernel"]                       # [K, Cin, Cout]
        y = F.conv1d(x, k.permute(2, 1, 0), p[conv(j) + "/bias"], padding=(k.shape[0] - 1) // 2)
        inv = p[bn(j) + "/gamma"] * torch.rsqrt(p[bn(j) + "/moving_variance"] + 1e-3)
        return y * inv[None, :, None] + (p[bn(j) + "/beta"] - p[bn(j) + "/moving_mean"] * inv)[None, :, None]

    def logits(self, x, keep_prob=1.0, generator=None, engine=None, masks=None):
        """x [N, 35] or [N, 35, 1] -> logits [N, 35].

        With ``engine`` (a fp32 HipEngine on the same GPU) the biGRU layers run on the native HIP
        forward/backward kernels (catfish_amd/native_train.py); everything else stays torch autograd.
        ``masks`` ({(layer, "fw"|"bw"): 0/1 array [N, 35, 64]}) replaces the random dropout draws (tests)."""
        torch = self.torch
        p = self.params
        x = torch.as_tensor(x, dtype=self.dtype, device=self.device)
        if x.dim() == 3:
            x = x[:, :, 0]
        a = x[:, None, :]                                # [N, C, T]
        hsz0 = int(p["stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/gru_cell/candidate/bias"].shape[0])
        csz0 = int(p["conv1d/bias"].shape[0]) if self.n_layers_res > 0 else 32
        anysize = engine is not None and not (hsz0 == 64 and csz0 == 32)     # recurrence on the any-size HIP kernels
        if anysize:
            return self._logits_anysize(a, keep_prob, generator, engine, masks)
        native_res = engine is not None and self.n_layers_res > 0 and self.dtype == torch.float32
        if native_res:                                   # residual conv stack on the HIP training kernels
            from .native_train import native_res_stack, res_unit_names
            a = native_res_stack(x, [p[k] for unit in res_unit_names(self.n_layers_res) for k in unit], engine)
        for d in range(0 if native_res else self.n_layers_res):
            j0 = 4 * d
            sc = self._conv_bn(a, j0)
            o = torch.relu(self._conv_bn(a, j0 + 1))
            o = torch.relu(self._conv_bn(o, j0 + 2))
            o = torch.relu(self._conv_bn(o, j0 + 3))
            a = torch.relu(o + sc)
        if not native_res:
            a = a.permute(0, 2, 1)                       # [N, T, C]
        n, t_len, _ = a.shape
        if engine is not None:                           # all biGRU layers on the HIP training kernels
            from .native_train import native_gru_stack
            pre = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell"
            plist = [p[(pre % (layer, d)) + k] for layer in range(self.n_layers) for d in ("fw", "bw")
                     for k in ("/gates/kernel", "/gates/bias", "/candidate/kernel", "/candidate/bias")]
            a = native_gru_stack(a, plist, engine, keep_prob, masks)
        for layer in range(self.n_layers if engine is None else 0):
            outs = []
            for dname, rev in (("fw", False), ("bw", True)):
                pre = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell" % (layer, dname)
                wg, bg = p[pre + "/gates/kernel"], p[pre + "/gates/bias"]
                wc, bc = p[pre + "/candidate/kernel"], p[pre + "/candidate/bias"]
                hsz = wc.shape[1]
                h = torch.zeros(n, hsz, dtype=self.dtype, device=self.device)
                seq = [None] * t_len
                for s in (range(t_len - 1, -1, -1) if rev else range(t_len)):
                    xt = a[:, s, :]
                    g = torch.sigmoid(torch.cat([xt, h], 1) @ wg + bg)
                    r, u = g[:, :hsz], g[:, hsz:]
                    c = torch.tanh(torch.cat([xt, r * h], 1) @ wc + bc)
                    h = u * h + (1 - u) * c
                    seq[s] = h
                out = torch.stack(seq, 1)
                if keep_prob < 1.0:                      # DropoutWrapper(output_keep_prob): outputs only
                    if masks is not None:
                        mask = torch.as_tensor(masks[(layer, dname)], dtype=self.dtype, device=self.device)
                    else:
                        mask = torch.floor(keep_prob + torch.rand(out.shape, generator=generator, device=self.device,
                                                                  dtype=self.dtype))
                    out = out / keep_prob * mask
                outs.append(out)
            a = torch.cat(outs, 2)
        return (a.reshape(-1, a.shape[2]) @ p["final_fully_connected/kernel"] +
                p["final_fully_connected/bias"]).reshape(n, t_len)

    def _logits_anysize(self, a, keep_prob, generator, engine, masks):
        """Any geometry other than 64 / 32: conv stack, dropout and head in torch, every biGRU layer one forward and one
        backward launch of the any-size HIP kernels (catfish_amd/anysize_train.py)."""
        torch = self.torch
        p = self.params
        from .anysize_train import anysize_bigru
        for d in range(self.n_layers_res):
            j0 = 4 * d
            sc = self._conv_bn(a, j0)
            o = torch.relu(self._conv_bn(a, j0 + 1))
            o = torch.relu(self._conv_bn(o, j0 + 2))
            o = torch.relu(self._conv_bn(o, j0 + 3))
            a = torch.relu(o + sc)
        a = a.permute(0, 2, 1).contiguous()              # [N, T, C]
        n, t_len, _ = a.shape
        pre = "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell"
        for layer in range(self.n_layers):
            params8 = [p[(pre % (layer, d)) + k] for d in ("fw", "bw")
                       for k in ("/gates/kernel", "/gates/bias", "/candidate/kernel", "/candidate/bias")]
            a = anysize_bigru(a, params8, engine)
            if keep_prob < 1.0:                          # DropoutWrapper(output_keep_prob), one draw per direction as in the torch path
                hsz = a.shape[2] // 2
                parts = []
                for dname in ("fw", "bw"):
                    if masks is not None:
                        parts.append(torch.as_tensor(masks[(layer, dname)], dtype=self.dtype, device=self.device))
                    else:
                        parts.append(torch.floor(keep_prob + torch.rand((n, t_len, hsz), generator=generator, device=self.device,
                                                                        dtype=self.dtype)))
                a = a / keep_prob * torch.cat(parts, 2)
        return (a.reshape(-1, a.shape[2]) @ p["final_fully_connected/kernel"] + p["final_fully_connected/bias"]).reshape(n, t_len)

    def loss(self, x, y, keep_prob=1.0, generator=None, engine=None, masks=None):
        """tf.losses.sigmoid_cross_entropy + reduce_mean (rnn_class.py:74-79)."""
        torch = self.torch
        z = self.logits(x, keep_prob, generator, engine, masks)
        y = torch.as_tensor(y, dtype=self.dtype, device=self.device).reshape(z.shape)
        return torch.nn.functional.binary_cross_entropy_with_logits(z, y, reduction="mean")


class TFOptimizer(object):
    """tf.train.AdamOptimizer / RMSPropOptimizer update rules with TF-1 defaults (rnn_class.py:62-71)."""

    def __init__(self, params, choice, lr):
        import torch
        self.torch = torch
        if choice not in ("Adam", "RMSProp"):
            raise ValueError("Given optimizer choice is not known. Choose 'Adam' or 'RMSProp'.")
        self.choice = choice
        self.lr = float(lr)
        self.keep_grads = False      # graph mode: gradients live in the captured pool and are overwritten on replay
        self.params = params
        first = next(iter(params.values()))
        # step counter as a DEVICE tensor so that the bias correction is computed by captured ops (hipGraph replay)
        self.t = torch.zeros((), dtype=torch.float64, device=first.device)
        self._b1 = torch.tensor(0.9, dtype=torch.float64, device=first.device)      # created outside any capture
        self._b2 = torch.tensor(0.999, dtype=torch.float64, device=first.device)
        if choice == "Adam":
            self.m = {k: torch.zeros_like(v) for k, v in params.items()}
            self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        else:
            self.ms = {k: torch.ones_like(v) for k, v in params.items()}      # TF initialises the rms slot to 1
            self.mom = {k: torch.zeros_like(v) for k, v in params.items()}

    # ---- slot variables under their TensorFlow names (what tf.train.Saver writes next to the weights) ----------
    def state_tf(self):
        """{checkpoint name: ndarray}: ``<var>/RMSProp`` (rms), ``<var>/RMSProp_1`` (momentum) or ``<var>/Adam`` (m),
        ``<var>/Adam_1`` (v), ``optimizer/beta{1,2}_power`` (the optimizer is built under name_scope "optimizer",
        rnn_class.py:62-71)."""
        out = {}
        if self.choice == "Adam":
            t = float(self.t)
            for k in self.params:
                out[k + "/Adam"] = self.m[k].detach().cpu().numpy().astype(np.float32)
                out[k + "/Adam_1"] = self.v[k].detach().cpu().numpy().astype(np.float32)
            out["optimizer/beta1_power"] = np.float32(0.9 ** (t + 1))       # TF stores beta^(t+1): initial value beta
            out["optimizer/beta2_power"] = np.float32(0.999 ** (t + 1))
        else:
            for k in self.params:
                out[k + "/RMSProp"] = self.ms[k].detach().cpu().numpy().astype(np.float32)
                out[k + "/RMSProp_1"] = self.mom[k].detach().cpu().numpy().astype(np.float32)
        return out

    def load_state_tf(self, state):
        """Continue from restored slot variables (``saver.restore`` brings them back, rnn_class.py:191-198).
        Entries of the other optimizer, or missing ones, leave the fresh initial value in place."""
        torch = self.torch
        a, b = ("Adam", "Adam_1") if self.choice == "Adam" else ("RMSProp", "RMSProp_1")
        first, second = (self.m, self.v) if self.choice == "Adam" else (self.ms, self.mom)
        with torch.no_grad():
            for k, p in self.params.items():
                for slot, name in ((first, k + "/" + a), (second, k + "/" + b)):
                    if name in state:
                        slot[k].copy_(torch.as_tensor(np.asarray(state[name]), dtype=p.dtype).reshape(p.shape))
            if self.choice == "Adam":
                # TF stores beta^(t+1).  beta1_power = 0.9^(t+1) underflows fp32 after ~830 steps (denormal, then 0), so
                # the step count is read from beta2_power = 0.999^(t+1) (good for ~87 000 steps); when that is gone too
                # the bias corrections are 1 to fp32 precision anyway and a large count stands in.
                t_new = None
                for name, beta in (("optimizer/beta2_power", 0.999), ("beta2_power", 0.999),
                                   ("optimizer/beta1_power", 0.9), ("beta1_power", 0.9)):
                    if name in state:
                        v = float(state[name])
                        if 1e-30 < v < 1.0:
                            t_new = round(np.log(v) / np.log(beta)) - 1
                            break
                        if v <= 1e-30:
                            t_new = 1000000
                if t_new is not None:
                    self.t.fill_(max(0, t_new))

    def step(self):
        """One update of every parameter with multi-tensor (``torch._foreach_*``) ops: a handful of launches
        instead of ~8 per parameter tensor (66 tensors)."""
        torch = self.torch
        with torch.no_grad():
            keys = [k for k, p in self.params.items() if p.grad is not None]
            ps = [self.params[k] for k in keys]
            gs = [self.params[k].grad for k in keys]
            self.t.add_(1.0)
            if self.choice == "Adam":
                b1, b2, eps = 0.9, 0.999, 1e-8
                lr_t = (self.lr * torch.sqrt(1.0 - torch.pow(self._b2, self.t)) / (1.0 - torch.pow(self._b1, self.t))).to(ps[0].dtype)
                ms = [self.m[k] for k in keys]
                vs = [self.v[k] for k in keys]
                torch._foreach_mul_(ms, b1)
                torch._foreach_add_(ms, gs, alpha=1 - b1)
                torch._foreach_mul_(vs, b2)
                torch._foreach_addcmul_(vs, gs, gs, value=1 - b2)
                den = torch._foreach_sqrt(vs)
                torch._foreach_add_(den, eps)
                upd = torch._foreach_div(ms, den)
                torch._foreach_mul_(upd, lr_t)                     # lr_t is a device scalar tensor: no host sync
                torch._foreach_sub_(ps, upd)
            else:
                decay, eps = 0.9, 1e-10
                rms = [self.ms[k] for k in keys]
                torch._foreach_mul_(rms, decay)
                torch._foreach_addcmul_(rms, gs, gs, value=1 - decay)
                den = torch._foreach_add(rms, eps)
                torch._foreach_sqrt_(den)
                upd = torch._foreach_div(gs, den)                  # momentum = 0: mom = lr * g / sqrt(ms + eps)
                torch._foreach_mul_(upd, self.lr)
                torch._foreach_sub_(ps, upd)
                torch._foreach_copy_([self.mom[k] for k in keys], upd)
            if not self.keep_grads:
                for p in ps:
                    p.grad = None


class Trainer(object):
    """One training step = forward + backward + optimizer update.

    On a GPU with the ResNetRNN type the step runs entirely on the HIP training kernels through the C ABI
    (``catfish_amd/native_step.py``: conv stack, biGRU layers, dense head + loss, optimizer; no autograd), captured ONCE
    into a HIP graph (``torch.cuda.CUDAGraph``) and replayed per batch; batches are copied into static input buffers.
    Dropout masks inside the graph come from torch's default (graph-safe) CUDA generator.  ``native=False`` (and the CPU)
    run the torch-autograd restatement of the same graph -- the reference the native step is tested against.
    """

    def __init__(self, weights, n_layers, n_layers_res, optimizer_choice, learning_rate, keep_prob, device=None,
                 seed=None, use_graph=None, native=None, optimizer_state=None, dtype=None):
        import torch
        if device is None:
            if not torch.cuda.is_available():
                # no silent CPU fallback: the torch-CPU mode exists as the reference for tests and is opt-in
                raise RuntimeError("Trainer: no HIP device visible; pass device='cpu' explicitly for the torch reference mode")
            device = "cuda"
        self.net = TorchResNetRNN(weights, n_layers, n_layers_res, device=device, dtype=dtype)
        self.opt = TFOptimizer(self.net.trainable(), optimizer_choice, learning_rate)
        if optimizer_state:
            self.opt.load_state_tf(optimizer_state)
        self.keep_prob = float(keep_prob)
        self.gen = torch.Generator(device=self.net.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))
        self.last_loss = None
        # native = the whole step on the HIP training kernels, no autograd: the default on a GPU for the ResNetRNN type at
        # the shipped geometry (64 GRU units, 32 conv channels), which is what those kernels are specialised for.  Other
        # draws of train_validate.generate_random_hyperparameters train through torch autograd with the biGRU recurrence
        # on the any-size HIP kernels (anysize_train.py); their inference runs on csrc/generic.hpp either way.
        h = int(np.asarray(weights["stack_bidirectional_rnn/cell_0/bidirectional_rnn/fw/gru_cell/candidate/bias"]).shape[0])
        c = int(np.asarray(weights["conv1d/bias"]).shape[0]) if n_layers_res > 0 else 32
        shipped = (h == 64 and c == 32)
        if native and not shipped:
            raise ValueError("the native training kernels are built for layer_size = 64 and layer_size_res = 32")
        self.native = (self.net.device.type == "cuda" and n_layers_res > 0 and shipped) if native is None else bool(native)
        self.engine = None
        self.step_impl = None
        # other geometries on a GPU: torch autograd around the any-size HIP recurrence kernels (anysize_train.py); the
        # engine is only the C-ABI handle those launches go through.  native=False keeps pure torch (the test reference).
        self.anysize = self.net.device.type == "cuda" and not shipped and native is None and self.net.dtype == torch.float32
        if self.anysize:
            from .engine import HipEngine
            self.engine = HipEngine(weights, layer_size=h, n_layers=n_layers, layer_size_res=c, n_layers_res=n_layers_res,
                                    device=self.net.device.index or 0, max_windows_per_pass=256)
        if self.native:
            from .engine import HipEngine
            self.engine = HipEngine(weights, n_layers=n_layers, n_layers_res=n_layers_res,
                                    device=self.net.device.index or 0, max_windows_per_pass=256, fuse_layers=False)
            if n_layers_res > 0 and self.net.dtype == torch.float32:
                from .native_step import NativeTrainStep
                self.step_impl = NativeTrainStep(self.net, self.opt, self.engine, self.keep_prob, seed=seed)
        self.use_graph = (self.net.device.type == "cuda") if use_graph is None else bool(use_graph)
        self._graph = None
        self._static = None
        if seed is not None and self.net.device.type == "cuda":
            torch.cuda.manual_seed(int(seed))

    def _capture(self, x, y):
        torch = self.torch_mod = __import__("torch")
        shape = tuple(np.asarray(x).reshape(-1, 35).shape)
        side = torch.cuda.Stream(self.net.device)
        side.wait_stream(torch.cuda.current_stream(self.net.device))
        if self.step_impl is not None:
            # native step: every launch goes through the C ABI on the capture stream; buffers are static per batch size
            b = self.step_impl._alloc(shape[0])
            with torch.cuda.stream(side):                   # warm-up off the capture stream (allocator, lazy init); no update
                for _ in range(2):
                    self.step_impl.run(b, update=False)
            torch.cuda.current_stream(self.net.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sloss = self.step_impl.run(b)
            self._graph, self._static = g, (b["x"], b, sloss)
            return
        sx = torch.zeros(shape, dtype=self.net.dtype, device=self.net.device)
        sy = torch.zeros_like(sx)
        self.opt.keep_grads = True
        with torch.cuda.stream(side):                       # warm-up off the capture stream (allocator, lazy init)
            for _ in range(2):
                loss = self.net.loss(sx, sy, self.keep_prob, None, self.engine)
                loss.backward()
                for p in self.net.trainable().values():
                    p.grad = None
        torch.cuda.current_stream(self.net.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sloss = self.net.loss(sx, sy, self.keep_prob, None, self.engine)
            sloss.backward()
            self.opt.step()
        self._graph, self._static = g, (sx, sy, sloss)

    def train_step(self, x, y, keep_prob=None, masks=None):
        kp = self.keep_prob if keep_prob is None else keep_prob
        if self.use_graph and kp == self.keep_prob and masks is None:
            torch = __import__("torch")
            if self._graph is None or tuple(self._static[0].shape) != tuple(np.asarray(x).reshape(-1, 35).shape):
                self._capture(x, y)          # first batch, or a new batch size: (re)capture the step
            sx, sy, sloss = self._static
            if self.step_impl is not None:
                self.step_impl.load_batch(sy, x, y)
            else:
                sx.copy_(torch.as_tensor(np.asarray(x), dtype=self.net.dtype).reshape(sx.shape), non_blocking=True)
                sy.copy_(torch.as_tensor(np.asarray(y), dtype=self.net.dtype).reshape(sy.shape), non_blocking=True)
            self._graph.replay()
            self.last_loss = float(sloss.detach())
            return self.last_loss
        if self.step_impl is not None:
            n = int(np.asarray(x).reshape(-1, 35).shape[0])
            if getattr(self, "_eager_bufs", None) is None or self._eager_bufs["n"] != n:
                self._eager_bufs = self.step_impl._alloc(n)
            self.step_impl.load_batch(self._eager_bufs, x, y)
            self.last_loss = float(self.step_impl.run(self._eager_bufs, keep_prob=kp, masks=masks))
            return self.last_loss
        loss = self.net.loss(x, y, kp, self.gen, self.engine, masks)
        loss.backward()
        self.opt.step()
        self.last_loss = float(loss.detach())
        return self.last_loss

    def gradients(self, x, y, keep_prob=None, masks=None):
        """Loss and gradients of one batch WITHOUT an update: (loss, {TF name: ndarray}) -- what tests compare with the
        reference graph's gradient tensors."""
        kp = self.keep_prob if keep_prob is None else keep_prob
        if self.step_impl is not None:
            n = int(np.asarray(x).reshape(-1, 35).shape[0])
            b = self.step_impl._alloc(n)
            self.step_impl.load_batch(b, x, y)
            loss = float(self.step_impl.run(b, keep_prob=kp, masks=masks, update=False))
            return loss, {k: v.detach().cpu().numpy().copy() for k, v in self.step_impl.grads().items()}
        for p in self.net.trainable().values():
            p.grad = None
        loss = self.net.loss(x, y, kp, self.gen, self.engine, masks)
        loss.backward()
        out = {k: p.grad.detach().cpu().numpy().copy() for k, p in self.net.trainable().items()}
        for p in self.net.trainable().values():
            p.grad = None
        return float(loss.detach()), out
