"""Training of biGRU layers of ANY geometry (layer_size 16..256, the draws of networks/train_validate.py:66-111).

The native training step (``native_step.py``) is built for the shipped 64 / 32 geometry.  For every other one the
recurrence -- in eager torch thousands of tiny kernels per step -- runs on the any-size HIP kernels through the C ABI
(``cf_gru_anysize_train_forward``: the inference kernel of ``csrc/generic.hpp`` plus a stash of the activated gates;
``cf_gru_anysize_train_backward``: backpropagation through the 35 steps), wrapped in a ``torch.autograd.Function``.
What is a plain GEMM over all (window, step) pairs stays a library GEMM: the input gradient ``da W_x^T`` and the weight
gradients ``[x; h]^T da``.  The weights are re-tiled on the device every step with gather maps computed here (numpy,
once per geometry), so no host round trip.  Residual blocks, dropout, dense head, loss and optimizer stay in torch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .native_train import T, frag_to_nat, nat_to_frag

LOG2E = 1.4426950408889634
GATE_SCALE = -LOG2E          # sigmoid(a) = 1 / (1 + exp2(-a log2 e))          (CF_GATE_SCALE)
CAND_SCALE = 2.0 * LOG2E     # tanh(a)    = 1 - 2 / (1 + exp2(2 a log2 e))      (CF_CAND_SCALE)

_MAPS = {}


def pack_maps(h, cin, device):
    """Gather maps of one direction: packed = src[idx] * scale with
    src = [gates_kernel (cin+h, 2h) | candidate_kernel (cin+h, h) | gates_bias (2h) | candidate_bias (h) | 0.0].

    Returns (w_idx, w_scale, b_idx, b_scale, wt_idx) device tensors in the layouts include/catfish_hip.h documents for
    cf_gru_anysize_train_forward / _backward (A-fragment order of csrc/generic.hpp: component i of lane l of block
    (mo, kb) = W[in = 16 kb + 4 (l >> 4) + i][out = 16 mo + (l & 15)])."""
    import torch
    key = (int(h), int(cin), str(device))
    if key in _MAPS:
        return _MAPS[key]
    h16, kbx = h // 16, (cin + 15) // 16
    kb_all = kbx + h16
    off_wc = (cin + h) * 2 * h
    off_bg = off_wc + (cin + h) * h
    off_bc = off_bg + 2 * h
    zero = off_bc + h
    lane = np.arange(64)
    q, n = lane >> 4, lane & 15

    def blocks(m16, k16):
        mo = np.arange(m16)[:, None, None, None]
        kb = np.arange(k16)[None, :, None, None]
        inn = 16 * kb + 4 * q[None, None, :, None] + np.arange(4)[None, None, None, :]
        out = 16 * mo + n[None, None, :, None] + 0 * inn
        return inn + 0 * out, out

    # forward: K = [x blocks (zero-padded to 16 kbx) | h blocks]
    inn, out = blocks(h16, kb_all)
    row = np.where(inn < 16 * kbx, np.where(inn < cin, inn, -1), cin + inn - 16 * kbx)
    idx_r = np.where(row >= 0, row * 2 * h + out, zero)
    idx_u = np.where(row >= 0, row * 2 * h + h + out, zero)
    idx_c = np.where(row >= 0, off_wc + row * h + out, zero)
    w_idx = np.stack([idx_r, idx_u, idx_c]).reshape(-1)
    w_scale = np.repeat(np.array([GATE_SCALE, GATE_SCALE, CAND_SCALE], np.float32), idx_r.size)
    f = (16 * np.arange(h16)[:, None, None] + 4 * q[None, :, None] + np.arange(4)[None, None, :])
    b_idx = np.stack([off_bg + f, off_bg + h + f, off_bc + f]).reshape(-1)
    b_scale = np.repeat(np.array([GATE_SCALE, GATE_SCALE, CAND_SCALE], np.float32), f.size)
    # backward: Wc_h^T (gate index in, state index out) and Wg_h^T
    inn, out = blocks(h16, h16)
    wct = off_wc + (cin + out) * h + inn
    inn, out = blocks(h16, 2 * h16)
    wgt = (cin + out) * 2 * h + inn
    wt_idx = np.concatenate([wct.reshape(-1), wgt.reshape(-1)])
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device)
    _MAPS[key] = (to(w_idx, np.int64), to(w_scale, np.float32), to(b_idx, np.int64), to(b_scale, np.float32), to(wt_idx, np.int64))
    return _MAPS[key]


def _make_function():
    import torch

    class AnySizeBiGRU(torch.autograd.Function):
        """y = biGRU_layer(x); x [N, 35, Cin] -> y [N, 35, 2H] (forward direction features first)."""

        @staticmethod
        def forward(ctx, x, wg_f, bg_f, wc_f, bc_f, wg_b, bg_b, wc_b, bc_b, engine):
            lib, handle = engine._lib, engine._handle
            n, _, cin = x.shape
            h = int(wc_f.shape[1])
            h16, kbx = h // 16, (cin + 15) // 16
            w_idx, w_scale, b_idx, b_scale, wt_idx = pack_maps(h, cin, x.device)
            dirs = ((wg_f, bg_f, wc_f, bc_f), (wg_b, bg_b, wc_b, bc_b))
            with torch.no_grad():
                zero = x.new_zeros(1, dtype=torch.float32)
                srcs = [torch.cat([wg.reshape(-1), wc.reshape(-1), bg.reshape(-1), bc.reshape(-1), zero]).float() for wg, bg, wc, bc in dirs]
                wpack = torch.stack([s[w_idx] * w_scale for s in srcs]).contiguous()
                bpack = torch.stack([s[b_idx] * b_scale for s in srcs]).contiguous()
                wtpack = torch.stack([s[wt_idx] for s in srcs]).contiguous()
                npad = (n + 15) // 16 * 16
                xp = x.float()
                if npad != n or 16 * kbx != cin:                # whole tiles of 16 windows, whole blocks of 16 features
                    xp = torch.nn.functional.pad(xp, (0, 16 * kbx - cin, 0, 0, 0, npad - n))
                x_frag = nat_to_frag(xp)
                tiles = npad // 16
                y_frag = torch.empty(tiles, T, 2 * h16, 64, 4, dtype=torch.float32, device=x.device)
                stash = torch.empty(tiles, T, 2, 3, h16, 64, 4, dtype=torch.float32, device=x.device)
                stream = torch.cuda.current_stream(x.device).cuda_stream
                N.check(lib.cf_gru_anysize_train_forward(handle, h, kbx, C.c_void_p(wpack.data_ptr()), C.c_void_p(bpack.data_ptr()),
                                                         C.c_void_p(x_frag.data_ptr()), C.c_void_p(y_frag.data_ptr()),
                                                         C.c_void_p(stash.data_ptr()), npad, C.c_void_p(stream)))
                y = frag_to_nat(y_frag)
            ctx.engine, ctx.n, ctx.npad, ctx.h, ctx.cin = engine, n, npad, h, cin
            ctx.save_for_backward(xp, y, y_frag, stash, wtpack, wg_f, wc_f, wg_b, wc_b)
            return y[:n].to(x.dtype)

        @staticmethod
        def backward(ctx, dy):
            xp, y, y_frag, stash, wtpack, wg_f, wc_f, wg_b, wc_b = ctx.saved_tensors
            engine, n, npad, h, cin = ctx.engine, ctx.n, ctx.npad, ctx.h, ctx.cin
            lib, handle = engine._lib, engine._handle
            dev = xp.device
            dyp = dy.float()
            if npad != n:
                dyp = torch.nn.functional.pad(dyp, (0, 0, 0, 0, 0, npad - n))
            dy_frag = nat_to_frag(dyp.contiguous())
            da = torch.empty_like(stash)
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(lib.cf_gru_anysize_train_backward(handle, h, C.c_void_p(wtpack.data_ptr()), C.c_void_p(y_frag.data_ptr()),
                                                      C.c_void_p(stash.data_ptr()), C.c_void_p(dy_frag.data_ptr()),
                                                      C.c_void_p(da.data_ptr()), npad, C.c_void_p(stream)))
            x2 = xp[:, :, :cin].reshape(npad * T, cin)
            dx = None
            grads = []
            for d, (wg, wc) in enumerate(((wg_f, wc_f), (wg_b, wc_b))):
                da_r, da_u, da_c = (frag_to_nat(da[:, :, d, g]) for g in range(3))          # [npad, 35, h] each
                r = frag_to_nat(stash[:, :, d, 0])
                hd = y[:, :, h * d:h * d + h]
                hprev = torch.zeros_like(hd)
                if d == 0:
                    hprev[:, 1:] = hd[:, :-1]                      # forward direction: the state before step t is y[t-1]
                else:
                    hprev[:, :-1] = hd[:, 1:]                      # backward direction: the state before step t is y[t+1]
                da_g = torch.cat([da_r, da_u], 2).reshape(npad * T, 2 * h)
                da_c = da_c.reshape(npad * T, h)
                a_g = torch.cat([x2, hprev.reshape(npad * T, h)], 1)
                a_c = torch.cat([x2, (r * hprev).reshape(npad * T, h)], 1)
                grads += [a_g.t() @ da_g, da_g.sum(0), a_c.t() @ da_c, da_c.sum(0)]
                dxd = da_g @ wg[:cin].float().t() + da_c @ wc[:cin].float().t()
                dx = dxd if dx is None else dx + dxd
            dx = dx.reshape(npad, T, cin)[:n].to(dy.dtype)
            return (dx,) + tuple(grads) + (None,)

    return AnySizeBiGRU


_FN = None


def anysize_bigru(x, params8, engine):
    """Differentiable biGRU layer of any geometry.  params8 = (wg_f, bg_f, wc_f, bc_f, wg_b, bg_b, wc_b, bc_b)."""
    global _FN
    if _FN is None:
        _FN = _make_function()
    return _FN.apply(x, *params8, engine)
