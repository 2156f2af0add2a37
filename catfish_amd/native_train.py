"""Native (HIP) forward + backward of the bidirectional GRU layers for training (BASELINE config 5).

The recurrent part is 95 % of the step's FLOPs and, in eager torch, thousands of tiny kernels.  Here each
biGRU layer is ONE forward launch (``cf_gru_train_forward``: the inference kernel plus a stash of the
activated gates) and ONE backward launch (``cf_gru_train_backward``: BPTT over the 35 steps with the
transposed-role weight fragments), wrapped in a ``torch.autograd.Function``.  The updated weights are
re-tiled on the device every step through the gather map of ``cf_gru_pack_map`` (no host round trip).  The weight gradients are
``A^T dA`` over all (window, step) pairs -- one library GEMM per matrix (torch.matmul -> rocBLAS); the
residual blocks, dropout masks, the dense head and the loss stay in torch autograd.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

T = 35


def nat_to_frag(x):
    """[N, 35, F] (N multiple of 16, F multiple of 16) -> fragment layout [N/16, 35, F/16, 64, 4]."""
    n, t, f = x.shape
    return x.reshape(n // 16, 16, t, f // 16, 4, 4).permute(0, 2, 3, 4, 1, 5).reshape(n // 16, t, f // 16, 64, 4).contiguous()


def frag_to_nat(x):
    """[tiles, 35, M, 64, 4] -> [tiles * 16, 35, 16 M]."""
    tiles, t, m = x.shape[0], x.shape[1], x.shape[2]
    return x.reshape(tiles, t, m, 4, 16, 4).permute(0, 4, 1, 2, 3, 5).reshape(tiles * 16, t, m * 16).contiguous()


_MAPS = {}


def pack_maps(cin, device):
    """(idx_fwd, scale_fwd, idx_bwd, scale_bwd) device tensors of cf_gru_pack_map, cached per (cin, device)."""
    import torch
    key = (int(cin), str(device))
    if key not in _MAPS:
        lib = N.lib()
        out = []
        for backward in (0, 1):
            n = C.c_int64()
            N.check(lib.cf_gru_pack_map(int(cin), backward, None, None, 0, C.byref(n)))
            idx = np.empty(n.value, dtype=np.int32)
            scale = np.empty(n.value, dtype=np.float32)
            N.check(lib.cf_gru_pack_map(int(cin), backward, idx.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p),
                                        n.value, C.byref(n)))
            out += [torch.from_numpy(idx.astype(np.int64)).to(device), torch.from_numpy(scale).to(device)]
        _MAPS[key] = tuple(out)
    return _MAPS[key]


def retile(params4, idx, scale):
    """One direction's (gates_kernel, gates_bias, candidate_kernel, candidate_bias) -> packed blob, on device."""
    import torch
    wg, bg, wc, bc = params4
    src = torch.cat([wg.reshape(-1), wc.reshape(-1), bg.reshape(-1), bc.reshape(-1), wg.new_zeros(1)]).float()
    return src[idx] * scale


def _make_function():
    import torch

    class NativeBiGRU(torch.autograd.Function):
        """y = biGRU_layer(x); x [N,35,Cin] -> y [N,35,128] (forward direction features first)."""

        @staticmethod
        def forward(ctx, x, wg_f, bg_f, wc_f, bc_f, wg_b, bg_b, wc_b, bc_b, engine):
            lib, handle = engine._lib, engine._handle
            cin = int(x.shape[2])
            idx_f, sc_f, idx_b, sc_b = pack_maps(cin, x.device)
            with torch.no_grad():
                dirs = ((wg_f, bg_f, wc_f, bc_f), (wg_b, bg_b, wc_b, bc_b))
                wpack = torch.stack([retile(d, idx_f, sc_f) for d in dirs]).contiguous()       # [2, n_floats] on device
                wpack_bwd = torch.stack([retile(d, idx_b, sc_b) for d in dirs]).contiguous()
            n = x.shape[0]
            npad = (n + 15) // 16 * 16
            xp = x if npad == n else torch.cat([x, x.new_zeros(npad - n, T, cin)], 0)
            x_frag = nat_to_frag(xp.float())
            tiles = npad // 16
            y_frag = torch.empty(tiles, T, 8, 64, 4, dtype=torch.float32, device=x.device)
            stash = torch.empty(tiles, T, 2, 12, 64, 4, dtype=torch.float32, device=x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            N.check(lib.cf_gru_train_forward(handle, cin, C.c_void_p(wpack.data_ptr()), C.c_void_p(x_frag.data_ptr()),
                                             C.c_void_p(y_frag.data_ptr()), C.c_void_p(stash.data_ptr()), npad, C.c_void_p(stream)))
            ctx.engine, ctx.n, ctx.npad = engine, n, npad
            ctx.save_for_backward(xp, y_frag, stash, wpack_bwd)
            return frag_to_nat(y_frag)[:n]

        @staticmethod
        def backward(ctx, dy):
            xp, y_frag, stash, wpack_bwd = ctx.saved_tensors
            engine, n, npad = ctx.engine, ctx.n, ctx.npad
            lib, handle = engine._lib, engine._handle
            dev = xp.device
            tiles = npad // 16
            cin = xp.shape[2]
            dyp = dy if npad == n else torch.cat([dy, dy.new_zeros(npad - n, T, dy.shape[2])], 0)
            dy_frag = nat_to_frag(dyp.float().contiguous())
            dx_frag = torch.empty(2, tiles, T, cin // 16, 64, 4, dtype=torch.float32, device=dev)
            da = torch.empty(tiles, T, 2, 12, 64, 4, dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(lib.cf_gru_train_backward(handle, int(cin), C.c_void_p(wpack_bwd.data_ptr()), C.c_void_p(y_frag.data_ptr()),
                                              C.c_void_p(stash.data_ptr()), C.c_void_p(dy_frag.data_ptr()), None, None,
                                              C.c_void_p(dx_frag.data_ptr()), C.c_void_p(da.data_ptr()), npad, C.c_void_p(stream)))
            dx = (frag_to_nat(dx_frag[0]) + frag_to_nat(dx_frag[1]))[:n]
            y = frag_to_nat(y_frag)                                   # [npad, 35, 128]
            grads = []
            x2 = xp.reshape(npad * T, cin)
            for d in range(2):
                da_d = frag_to_nat(da[:, :, d])                       # [npad, 35, 192]: da_r | da_u | da_c
                r_d = frag_to_nat(stash[:, :, d, 0:4])                # [npad, 35, 64]
                h = y[:, :, 64 * d:64 * d + 64]
                hprev = torch.zeros_like(h)
                if d == 0:
                    hprev[:, 1:] = h[:, :-1]                          # forward direction: state before step t is y[t-1]
                else:
                    hprev[:, :-1] = h[:, 1:]                          # backward direction: state before step t is y[t+1]
                da_g = da_d[:, :, :128].reshape(npad * T, 128)
                da_c = da_d[:, :, 128:].reshape(npad * T, 64)
                a_g = torch.cat([x2, hprev.reshape(npad * T, 64)], 1)
                a_c = torch.cat([x2, (r_d * hprev).reshape(npad * T, 64)], 1)
                grads += [a_g.t() @ da_g, da_g.sum(0), a_c.t() @ da_c, da_c.sum(0)]
            return (dx.to(dy.dtype),) + tuple(grads) + (None,)

    return NativeBiGRU


_FN = None


def native_bigru(x, params8, engine):
    """Differentiable biGRU layer on the HIP kernels.  params8 = (wg_f, bg_f, wc_f, bc_f, wg_b, bg_b, wc_b, bc_b)."""
    global _FN
    if _FN is None:
        _FN = _make_function()
    return _FN.apply(x, *params8, engine)


# ------------------------------------------------------------------------------------------------
# The whole biGRU stack, fragment-resident (what the Trainer uses)
# ------------------------------------------------------------------------------------------------
_STACK_MAPS = {}


def _stack_maps(cins, device):
    """One gather map for ALL layers, directions and both packings: packed = cat(blocks)[idx] * scale, where the
    source is the concatenation of per-(layer, direction) blocks [gates_kernel | candidate_kernel | gates_bias |
    candidate_bias | 0.0].  Returns (idx, scale, [(fwd_offset, fwd_floats, bwd_offset, bwd_floats) per layer])."""
    import torch
    key = (tuple(cins), str(device))
    if key not in _STACK_MAPS:
        idxs, scales, layout = [], [], []
        src_off = 0
        out_off = 0
        fwd_parts, bwd_parts = [], []
        for cin in cins:
            idx_f, sc_f, idx_b, sc_b = pack_maps(cin, device)
            block = (cin + 64) * 192 + 192 + 1                       # wg + wc + bg + bc + the zero slot
            fwd_parts.append([(idx_f + src_off + d * block, sc_f) for d in range(2)])
            bwd_parts.append([(idx_b + src_off + d * block, sc_b) for d in range(2)])
            src_off += 2 * block
        for parts in (fwd_parts, bwd_parts):
            offs = []
            for layer in parts:
                n = sum(int(i.numel()) for i, _ in layer)
                offs.append((out_off, n))
                for i, sc in layer:
                    idxs.append(i)
                    scales.append(sc)
                out_off += n
            layout.append(offs)
        per_layer = [(layout[0][k][0], layout[0][k][1], layout[1][k][0], layout[1][k][1]) for k in range(len(cins))]
        _STACK_MAPS[key] = (torch.cat(idxs), torch.cat(scales), per_layer)
    return _STACK_MAPS[key]


def _make_stack_function():
    import torch

    class NativeGRUStack(torch.autograd.Function):
        """y = dropout(biGRU_L(... dropout(biGRU_1(x)) ...)); x [N,35,Cin0] -> y [N,35,128].

        Activations stay in the MFMA fragment layout between the layers; per layer one forward launch, one
        backward launch (the two direction slabs of the layer above are added, and the dropout factor applied,
        inside it) and one weight-gradient launch.  ``drop``: None, or one [npad/16,35,8,64,4] tensor per layer
        holding mask / keep_prob in fragment layout (DropoutWrapper(output_keep_prob), rnn_class.py:151-154)."""

        @staticmethod
        def forward(ctx, x, engine, drop, *params):
            lib, handle = engine._lib, engine._handle
            n_layers = len(params) // 8
            dev = x.device
            cins = [int(x.shape[2])] + [128] * (n_layers - 1)
            idx, scale, per_layer = _stack_maps(cins, dev)
            with torch.no_grad():
                zero = x.new_zeros(1, dtype=torch.float32)
                blocks = []
                for layer in range(n_layers):
                    for d in range(2):
                        wg, bg, wc, bc = params[8 * layer + 4 * d:8 * layer + 4 * d + 4]
                        blocks += [wg.reshape(-1), wc.reshape(-1), bg.reshape(-1), bc.reshape(-1), zero]
                packed = torch.cat(blocks).float()[idx] * scale
            n = x.shape[0]
            npad = (n + 15) // 16 * 16
            tiles = npad // 16
            xp = x if npad == n else torch.cat([x, x.new_zeros(npad - n, T, cins[0])], 0)
            cur = nat_to_frag(xp.float())
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            saved = []
            for layer in range(n_layers):
                fo, fn, _, _ = per_layer[layer]
                y_frag = torch.empty(tiles, T, 8, 64, 4, dtype=torch.float32, device=dev)
                stash = torch.empty(tiles, T, 2, 12, 64, 4, dtype=torch.float32, device=dev)
                N.check(lib.cf_gru_train_forward(handle, cins[layer], C.c_void_p(packed[fo:fo + fn].data_ptr()),
                                                 C.c_void_p(cur.data_ptr()), C.c_void_p(y_frag.data_ptr()),
                                                 C.c_void_p(stash.data_ptr()), npad, stream))
                saved += [cur, y_frag, stash]
                cur = y_frag if drop is None else y_frag * drop[layer]
            ctx.engine, ctx.n, ctx.npad, ctx.cins, ctx.per_layer = engine, n, npad, cins, per_layer
            ctx.drop = drop
            ctx.save_for_backward(packed, *saved)
            return frag_to_nat(cur)[:n]

        @staticmethod
        def backward(ctx, dy):
            packed = ctx.saved_tensors[0]
            saved = ctx.saved_tensors[1:]
            engine, n, npad, cins, per_layer, drop = ctx.engine, ctx.n, ctx.npad, ctx.cins, ctx.per_layer, ctx.drop
            lib, handle = engine._lib, engine._handle
            dev = dy.device
            tiles = npad // 16
            n_layers = len(cins)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            dyp = dy if npad == n else torch.cat([dy, dy.new_zeros(npad - n, T, dy.shape[2])], 0)
            g0 = nat_to_frag(dyp.float().contiguous())
            g1 = None
            grads = [None] * (8 * n_layers)
            for layer in range(n_layers - 1, -1, -1):
                cin = cins[layer]
                x_frag, y_frag, stash = saved[3 * layer:3 * layer + 3]
                _, _, bo, bn = per_layer[layer]
                dx_frag = torch.empty(2, tiles, T, cin // 16, 64, 4, dtype=torch.float32, device=dev)
                da = torch.empty(tiles, T, 2, 12, 64, 4, dtype=torch.float32, device=dev)
                N.check(lib.cf_gru_train_backward(
                    handle, cin, C.c_void_p(packed[bo:bo + bn].data_ptr()), C.c_void_p(y_frag.data_ptr()), C.c_void_p(stash.data_ptr()),
                    C.c_void_p(g0.data_ptr()), None if g1 is None else C.c_void_p(g1.data_ptr()),
                    None if drop is None else C.c_void_p(drop[layer].data_ptr()),
                    C.c_void_p(dx_frag.data_ptr()), C.c_void_p(da.data_ptr()), npad, stream))
                ws_floats = int(lib.cf_gru_wgrad_workspace_floats(handle, cin, npad))
                ws = torch.empty(ws_floats, dtype=torch.float32, device=dev)
                rows = cin + 64
                out = torch.empty(2, rows * 192 + 192, dtype=torch.float32, device=dev)
                N.check(lib.cf_gru_train_wgrad(handle, cin, C.c_void_p(x_frag.data_ptr()), C.c_void_p(y_frag.data_ptr()),
                                               C.c_void_p(stash.data_ptr()), C.c_void_p(da.data_ptr()), npad,
                                               C.c_void_p(ws.data_ptr()), ws_floats, C.c_void_p(out.data_ptr()), stream))
                for d in range(2):
                    o = out[d]
                    grads[8 * layer + 4 * d + 0] = o[:rows * 128].view(rows, 128)
                    grads[8 * layer + 4 * d + 1] = o[rows * 128:rows * 128 + 128]
                    grads[8 * layer + 4 * d + 2] = o[rows * 128 + 128:rows * 128 + 128 + rows * 64].view(rows, 64)
                    grads[8 * layer + 4 * d + 3] = o[rows * 128 + 128 + rows * 64:]
                g0, g1 = dx_frag[0], dx_frag[1]
            dx = frag_to_nat(g0 + g1)[:n]
            return (dx.to(dy.dtype), None, None) + tuple(grads)

    return NativeGRUStack


_STACK_FN = None


def dropout_scale_frag(npad, keep_prob, device, masks=None, layer=None):
    """mask / keep_prob for one layer's [npad, 35, 128] output, in fragment layout.  ``masks`` (tests):
    {(layer, "fw"|"bw"): 0/1 array [n, 35, 64]} replaces the random draw."""
    import torch
    if masks is not None:
        m = torch.cat([torch.as_tensor(np.asarray(masks[(layer, d)]), dtype=torch.float32, device=device) for d in ("fw", "bw")], 2)
        if m.shape[0] < npad:
            m = torch.cat([m, m.new_zeros(npad - m.shape[0], T, 128)], 0)
        return nat_to_frag(m) / keep_prob
    return torch.floor(keep_prob + torch.rand(npad // 16, T, 8, 64, 4, device=device, dtype=torch.float32)) / keep_prob


def native_gru_stack(x, params, engine, keep_prob=1.0, masks=None):
    """Differentiable stack of biGRU layers on the HIP kernels.  params: 8 tensors per layer
    (wg_f, bg_f, wc_f, bc_f, wg_b, bg_b, wc_b, bc_b), x: [N, 35, Cin0] -> [N, 35, 128]."""
    global _STACK_FN
    if _STACK_FN is None:
        _STACK_FN = _make_stack_function()
    drop = None
    if keep_prob < 1.0:
        import torch
        n_layers = len(params) // 8
        npad = (int(x.shape[0]) + 15) // 16 * 16
        if masks is not None:
            drop = [dropout_scale_frag(npad, keep_prob, x.device, masks, layer) for layer in range(n_layers)]
        else:                                            # one draw for all layers: 3 launches instead of 3 per layer
            drop = torch.floor(keep_prob + torch.rand(n_layers, npad // 16, T, 8, 64, 4, device=x.device, dtype=torch.float32))
            drop = list((drop / keep_prob).unbind(0))
    return _STACK_FN.apply(x, engine, drop, *params)


# ------------------------------------------------------------------------------------------------
# The residual conv stack (resnet_class.py:44-82) on the HIP training kernels
# ------------------------------------------------------------------------------------------------
def res_unit_names(n_blocks):
    """TF variable names per conv+BN unit, in the order of the kernels' parameter buffer (include/catfish_hip.h)."""
    conv = lambda j: "conv1d" if j == 0 else "conv1d_%d" % j                                  # noqa: E731
    bn = lambda j: "batch_normalization" if j == 0 else "batch_normalization_%d" % j         # noqa: E731
    out = []
    for j in range(4 * n_blocks):
        out.append([conv(j) + "/kernel", conv(j) + "/bias", bn(j) + "/gamma", bn(j) + "/beta", bn(j) + "/moving_mean",
                    bn(j) + "/moving_variance"])
    return out


def _make_res_function():
    import torch

    class NativeResStack(torch.autograd.Function):
        """out = residual_blocks(x); x [N,35] -> [N,35,32].  params: 6 tensors per unit (res_unit_names order)."""

        @staticmethod
        def forward(ctx, x, engine, *params):
            lib, handle = engine._lib, engine._handle
            n_blocks = len(params) // 24
            dev = x.device
            n = int(x.shape[0])
            with torch.no_grad():
                prm = torch.cat([p.reshape(-1) for p in params]).float()
            assert prm.numel() == lib.cf_res_train_param_floats(n_blocks)
            xc = x.float().contiguous()
            z = torch.empty(4 * n_blocks, n * T, 32, dtype=torch.float32, device=dev)
            out = torch.empty(n, T, 32, dtype=torch.float32, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            N.check(lib.cf_res_train_forward(handle, n_blocks, C.c_void_p(prm.data_ptr()), C.c_void_p(xc.data_ptr()),
                                             C.c_void_p(z.data_ptr()), C.c_void_p(out.data_ptr()), n, stream))
            ctx.engine, ctx.n_blocks, ctx.shapes = engine, n_blocks, [tuple(p.shape) for p in params]
            ctx.save_for_backward(prm, xc, z)
            return out

        @staticmethod
        def backward(ctx, dout):
            prm, xc, z = ctx.saved_tensors
            engine, n_blocks = ctx.engine, ctx.n_blocks
            lib, handle = engine._lib, engine._handle
            dev = dout.device
            n = int(xc.shape[0])
            ws_floats = int(lib.cf_res_train_workspace_floats(n_blocks, n))
            ws = torch.empty(ws_floats, dtype=torch.float32, device=dev)
            grads = torch.empty_like(prm)
            d = dout.float().contiguous()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            N.check(lib.cf_res_train_backward(handle, n_blocks, C.c_void_p(prm.data_ptr()), C.c_void_p(xc.data_ptr()),
                                              C.c_void_p(z.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(ws.data_ptr()), ws_floats,
                                              C.c_void_p(grads.data_ptr()), n, stream))
            outs, off = [], 0
            for i, shp in enumerate(ctx.shapes):
                cnt = int(np.prod(shp))
                outs.append(None if i % 6 >= 4 else grads[off:off + cnt].view(shp))      # moving statistics do not train
                off += cnt
            return (None, None) + tuple(outs)

    return NativeResStack


_RES_FN = None


def native_res_stack(x, params, engine):
    """Differentiable residual conv stack on the HIP kernels: x [N,35] -> [N,35,32]."""
    global _RES_FN
    if _RES_FN is None:
        _RES_FN = _make_res_function()
    return _RES_FN.apply(x, engine, *params)
