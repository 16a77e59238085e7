"""Host-side mirror of the reference's layer surface (models/layers.py) on top of
``libfgnn_hip.so``.

Same class names, constructor signatures, attribute names (``.convs``, ``.gn``) and
parameter shapes as the reference, so ``state_dict``s are interchangeable
(``MlpBlock_Real`` :109-131, ``GraphNorm`` :47-69, ``normalize`` :71-80, ``Matmul``
:161-162, ``Concat`` :145-146, ``ColumnMaxPooling`` :194-203, ``Identity`` :151-152).
Each forward/backward is a HIP kernel sequence behind a ``torch.autograd.Function``;
inputs must live on the GPU -- there is no CPU path in this package.
A ``MaskedTensor`` input (ragged batch) is handled by passing its per-graph vertex
counts to the kernels; the result is again a ``MaskedTensor`` with exact zeros in the
padding.
"""
import ctypes as C
import math
import os
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import _lib
from .masked import MaskedTensor

FGNN_H = _lib.FGNN_H
FUSED_INPUT_WIDTHS = (2, 16, 32, 32 + 2, 32 + 32)      # fgnn_mlp_fwd / fgnn_mlp_bwd instantiations


def _split(x):
    """(tensor | MaskedTensor) -> (dense tensor, nvalid or None)."""
    if isinstance(x, MaskedTensor):
        return x.tensor, x.nvalid
    return x, None


def _wrap(y, like, masked_dims=None):
    if isinstance(like, MaskedTensor):
        return MaskedTensor(y, like.nvalid, masked_dims or like.masked_dims, like.base_name)
    return y


def _check(x, name):
    if not x.is_cuda:
        raise RuntimeError('%s: tensor is on %s; graph_neural_net_amd only runs on the GPU (no CPU fallback)'
                           % (name, x.device))
    if x.dtype != torch.float32:
        raise RuntimeError('%s: only float32 is supported (got %s)' % (name, x.dtype))


def _nv(nvalid):
    return _lib.ptr(nvalid) if nvalid is not None else None


def _off(t, floats):
    return C.c_void_p(t.data_ptr() + 4 * floats)


def _input_slabs(x, cin, P):
    """Describe a (G, cin, N, N) tensor as one or two <=32-channel slabs."""
    a = _lib.make_slab(x, cin * P, P, min(cin, 32))
    b = None
    if cin > 32:
        b = _lib.make_slab(x, cin * P, P, cin - 32)
        b.ptr = x.data_ptr() + 4 * 32 * P
    return a, b


# --------------------------------------------------------------------------------------
# MlpBlock_Real = conv1x1/ReLU chain + GraphNorm, one fused forward and one fused backward
# --------------------------------------------------------------------------------------
class _MlpGnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nvalid, eps, gn_w, gn_b, *wb):
        _check(x, 'MlpBlock_Real')
        x = x.contiguous()
        G, cin, N, _ = x.shape
        P = N * N
        depth = len(wb) // 2
        dev = x.device
        tpg = _lib.tiles_per_graph(N)
        z = torch.empty(G, FGNN_H, N, N, dtype=torch.float32, device=dev)
        y = torch.empty_like(z)
        part = torch.empty(G * tpg * FGNN_H * 2, dtype=torch.float32, device=dev)
        cnt = torch.empty(G * tpg, dtype=torch.float32, device=dev)
        nrm = torch.empty(G * FGNN_H * 4, dtype=torch.float32, device=dev)
        wb = [t.contiguous() for t in wb]
        args = _lib.MlpFwdArgs()
        args.G, args.N, args.depth, args.nmlp = G, N, depth, 1
        args.nvalid = nvalid.data_ptr() if nvalid is not None else None
        a, b = _input_slabs(x, cin, P)
        args.a = a
        if b is not None:
            args.b = b
        for l in range(depth):
            if wb[2 * l].shape[0] != FGNN_H:
                raise RuntimeError('MlpBlock_Real: the HIP kernels are built for out_features = 32')
            args.W[0][l] = wb[2 * l].data_ptr()
            args.bias[0][l] = wb[2 * l + 1].data_ptr()
        args.z[0] = z.data_ptr()
        args.part[0] = part.data_ptr()
        args.ldz = P
        args.cnt = cnt.data_ptr()
        st = _lib.stream_ptr()
        _lib.call('fgnn_mlp_fwd', C.byref(args), st)
        _lib.call('fgnn_gn_finalize', _lib.ptr(part), _lib.ptr(cnt), _lib.ptr(gn_w), _nv(nvalid), G, FGNN_H, N,
                  float(eps), _lib.ptr(nrm), st)
        _lib.call('fgnn_gn_apply', _lib.ptr(z), FGNN_H * P, P, _lib.ptr(nrm), _lib.ptr(gn_b), _nv(nvalid), G, FGNN_H, N,
                  _lib.ptr(y), FGNN_H * P, P, st)
        ctx.save_for_backward(x, z, nrm, nvalid, gn_w, gn_b, *wb)
        ctx.depth = depth
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, nrm, nvalid, gn_w, gn_b = ctx.saved_tensors[:6]
        wb = ctx.saved_tensors[6:]
        depth = ctx.depth
        dy = dy.contiguous()
        G, cin, N, _ = x.shape
        P = N * N
        dev = x.device
        st = _lib.stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        s12 = torch.empty(G * FGNN_H * 2, **f32)
        coef = torch.empty(G * FGNN_H * 4, **f32)
        dgw = torch.empty(FGNN_H, **f32) if gn_w is not None else None
        dgb = torch.empty(FGNN_H, **f32) if gn_b is not None else None
        gs = FGNN_H * P
        _lib.call('fgnn_gn_bwd_stats', _lib.ptr(dy), gs, P, _lib.ptr(z), gs, P, _lib.ptr(nrm), _nv(nvalid), G, FGNN_H, N,
                  _lib.ptr(s12), st)
        _lib.call('fgnn_gn_bwd_coef', _lib.ptr(s12), _lib.ptr(nrm), _nv(nvalid), G, FGNN_H, N, _lib.ptr(coef),
                  _lib.ptr(dgw), _lib.ptr(dgb), st)
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        nwg = _lib.load().fgnn_mlp_bwd_num_workgroups()
        pcount = _lib.mlp_param_count(cin, depth)
        wpart = torch.empty(nwg * pcount, **f32)
        flat = torch.empty(pcount, **f32)
        args = _lib.MlpBwdArgs()
        args.G, args.N, args.depth = G, N, depth
        args.nvalid = nvalid.data_ptr() if nvalid is not None else None
        a, b = _input_slabs(x, cin, P)
        args.a = a
        if b is not None:
            args.b = b
        for l in range(depth):
            args.W[l] = wb[2 * l].data_ptr()
            args.bias[l] = wb[2 * l + 1].data_ptr()
        args.dy, args.dgstride, args.ldd = dy.data_ptr(), gs, P
        args.z, args.zgstride, args.ldz = z.data_ptr(), gs, P
        args.coef = coef.data_ptr()
        if need_dx:
            args.dxa, args.dxa_gstride, args.dxa_ld = dx.data_ptr(), cin * P, P
            if cin > 32:
                args.dxb, args.dxb_gstride, args.dxb_ld = dx.data_ptr() + 4 * 32 * P, cin * P, P
        args.wpart = wpart.data_ptr()
        _lib.call('fgnn_mlp_bwd', C.byref(args), st)
        _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), nwg, pcount, _lib.ptr(flat), st)
        grads = []
        off = 0
        c = cin
        for l in range(depth):
            grads.append(flat[off:off + 32 * c].view(32, c, 1, 1))
            off += 32 * c
            grads.append(flat[off:off + 32])
            off += 32
            c = 32
        return (dx, None, None,
                dgw.view(1, FGNN_H, 1, 1) if dgw is not None else None,
                dgb.view(1, FGNN_H, 1, 1) if dgb is not None else None, *grads)


# --------------------------------------------------------------------------------------
# One 1x1 conv (+ ReLU) of any width: the building block of MlpBlock_Real outside the fused 32-wide kernels
# --------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """y = act(conv1x1(x)) (models/layers.py:125-131, one loop iteration) as csrc/conv.hip launches."""

    @staticmethod
    def forward(ctx, x, nvalid, weight, bias, relu):
        _check(x, 'MlpBlock_Real')
        x = x.contiguous()
        w = weight.contiguous()
        G, K, N, _ = x.shape
        M = w.shape[0]
        if w.shape[1] != K:
            raise RuntimeError('MlpBlock_Real: conv expects %d input channels, the tensor has %d' % (w.shape[1], K))
        P = N * N
        y = torch.empty(G, M, N, N, dtype=torch.float32, device=x.device)
        _lib.call('fgnn_conv1x1', _lib.ptr(x), K * P, P, None, _lib.ptr(w), K, 1, _lib.ptr(bias), int(relu), _nv(nvalid),
                  G, N, M, K, _lib.ptr(y), M * P, P, _lib.stream_ptr())
        ctx.save_for_backward(x, y if relu else None, w, nvalid)
        ctx.relu, ctx.has_bias = relu, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, nvalid = ctx.saved_tensors
        dy = dy.contiguous()
        G, K, N, _ = x.shape
        M = w.shape[0]
        P = N * N
        st = _lib.stream_ptr()
        f32 = dict(dtype=torch.float32, device=x.device)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call('fgnn_conv1x1', _lib.ptr(dy), M * P, P, _lib.ptr(y), _lib.ptr(w), 1, K, None, 0, _nv(nvalid),
                      G, N, K, M, _lib.ptr(dx), K * P, P, st)
        dw = db = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            chunks = _lib.load().fgnn_conv1x1_dw_chunks(G, N)
            cnt = M * K + M
            wpart = torch.empty(chunks * cnt, **f32)
            flat = torch.empty(cnt, **f32)
            _lib.call('fgnn_conv1x1_dw', _lib.ptr(dy), M * P, P, _lib.ptr(y), _lib.ptr(x), K * P, P, _nv(nvalid), G, N, M, K,
                      _lib.ptr(wpart), st)
            _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), chunks, cnt, _lib.ptr(flat), st)
            dw = flat[:M * K].view(M, K, 1, 1)
            db = flat[M * K:] if ctx.has_bias else None
        return dx, None, dw, db, None


def _chain_supported(cin, widths):
    arr = (C.c_int * len(widths))(*widths)
    return len(widths) <= 3 and bool(_lib.load().fgnn_conv_chain_supported(len(widths), cin, arr))


class _ConvChainFn(torch.autograd.Function):
    """The conv stack of MlpBlock_Real (models/layers.py:125-131: conv, ReLU, ..., conv) outside the fused 32-wide kernels as
    ONE launch forward (csrc/conv.hip `conv_chain_kernel`: the hidden activations stay in registers and are written once, for
    the backward) and ONE launch for the whole input-gradient chain; the weight gradients take the d(pre-activation) tensors
    that chain leaves behind (fgnn_conv1x1_dw, no mask to re-read)."""

    @staticmethod
    def forward(ctx, x, nvalid, *wb):
        _check(x, 'MlpBlock_Real')
        x = x.contiguous()
        G, K0, N, _ = x.shape
        P = N * N
        depth = len(wb) // 2
        ws = [wb[2 * l].contiguous() for l in range(depth)]
        bs = [wb[2 * l + 1] for l in range(depth)]
        acts = []
        args = _lib.ChainArgs()
        args.x, args.x_gstride, args.x_ld, args.depth = x.data_ptr(), K0 * P, P, depth
        args.o_ld, args.G, args.N = P, G, N
        args.nvalid = nvalid.data_ptr() if nvalid is not None else None
        k = K0
        for l in range(depth):
            M = ws[l].shape[0]
            if ws[l].shape[1] != k:
                raise RuntimeError('MlpBlock_Real: conv %d expects %d input channels, gets %d' % (l, ws[l].shape[1], k))
            a = torch.empty(G, M, N, N, dtype=torch.float32, device=x.device)
            acts.append(a)
            L = args.layer[l]
            L.W, L.w_ostride, L.w_kstride = ws[l].data_ptr(), k, 1
            L.bias = bs[l].data_ptr() if bs[l] is not None else None
            L.M, L.K, L.relu = M, k, int(l < depth - 1)
            L.mask, L.out, L.o_gstride = None, a.data_ptr(), M * P
            k = M
        _lib.call('fgnn_conv_chain', C.byref(args), _lib.stream_ptr())
        ctx.save_for_backward(x, nvalid, *acts, *ws)
        ctx.depth, ctx.has_bias = depth, [b is not None for b in bs]
        return acts[-1]

    @staticmethod
    def backward(ctx, dz):
        depth = ctx.depth
        x, nvalid = ctx.saved_tensors[:2]
        acts = ctx.saved_tensors[2:2 + depth]
        ws = ctx.saved_tensors[2 + depth:]
        dz = dz.contiguous()
        G, K0, N, _ = x.shape
        P = N * N
        st = _lib.stream_ptr()
        f32 = dict(dtype=torch.float32, device=x.device)
        need_dx = ctx.needs_input_grad[0]
        # input-gradient chain: layer j of the chain is conv (depth-1-j) transposed; its output is d(pre-activation) of conv
        # (depth-2-j) -- masked by that conv's saved post-ReLU output -- or, for the last one, dx
        # (a last chain layer of more than 64 outputs -- dx of a block's mlp3, whose input is the 128-channel cat -- needs four
        #  fragment groups per wave: one wave per SIMD, 238 us against 60 for the plain kernel; it stays a launch of its own)
        split_dx = need_dx and ws[0].shape[1] > 64
        nchain = depth if (need_dx and not split_dx) else depth - 1
        dpre = [None] * depth           # dpre[l] = gradient at the output of conv l (before its ReLU mask for l = depth-1: none)
        dpre[depth - 1] = dz
        dx = None
        if nchain > 0:
            args = _lib.ChainArgs()
            args.x, args.x_gstride, args.x_ld, args.depth = dz.data_ptr(), ws[depth - 1].shape[0] * P, P, nchain
            args.o_ld, args.G, args.N = P, G, N
            args.nvalid = nvalid.data_ptr() if nvalid is not None else None
            for j in range(nchain):
                l = depth - 1 - j                       # the conv being transposed: M_l outputs -> K_l inputs
                Ml, Kl = ws[l].shape[0], ws[l].shape[1]
                out = torch.empty(G, Kl, N, N, **f32)
                L = args.layer[j]
                L.W, L.w_ostride, L.w_kstride, L.bias = ws[l].data_ptr(), 1, Kl, None
                L.M, L.K, L.relu = Kl, Ml, 0
                L.mask = acts[l - 1].data_ptr() if l > 0 else None
                L.out, L.o_gstride = out.data_ptr(), Kl * P
                if l > 0:
                    dpre[l - 1] = out
                else:
                    dx = out
            _lib.call('fgnn_conv_chain', C.byref(args), st)
        if split_dx:
            M0, K0w = ws[0].shape[0], ws[0].shape[1]
            dx = torch.empty(G, K0w, N, N, **f32)
            _lib.call('fgnn_conv1x1', _lib.ptr(dpre[0]), M0 * P, P, None, _lib.ptr(ws[0]), 1, K0w, None, 0, _nv(nvalid), G, N, K0w, M0,
                      _lib.ptr(dx), K0w * P, P, st)
        # weight / bias gradients of every layer: one launch, one reduction (a chunk's record = the layers' records in a row)
        want = [ctx.needs_input_grad[2 + 2 * l] or (ctx.has_bias[l] and ctx.needs_input_grad[3 + 2 * l]) for l in range(depth)]
        grads = [None] * (2 * depth)
        todo = [l for l in range(depth) if want[l]]
        if todo:
            chunks = _lib.load().fgnn_conv1x1_dw_chunks(G, N)
            jobs = (_lib.DwJob * len(todo))()
            offs, total = [], 0
            for i, l in enumerate(todo):
                Ml, Kl = ws[l].shape[0], ws[l].shape[1]
                inp = acts[l - 1] if l > 0 else x
                jb = jobs[i]
                jb.dy, jb.d_gstride, jb.d_ld, jb.relu_mask = dpre[l].data_ptr(), Ml * P, P, None
                jb.x, jb.x_gstride, jb.x_ld, jb.M, jb.K = inp.data_ptr(), Kl * P, P, Ml, Kl
                offs.append(total)
                total += Ml * Kl + Ml
            wpart = torch.empty(chunks * total, **f32)
            flat = torch.empty(total, **f32)
            _lib.call('fgnn_conv1x1_dw_multi', jobs, len(todo), _nv(nvalid), G, N, _lib.ptr(wpart), st)
            _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), chunks, total, _lib.ptr(flat), st)
            for i, l in enumerate(todo):
                Ml, Kl = ws[l].shape[0], ws[l].shape[1]
                grads[2 * l] = flat[offs[i]:offs[i] + Ml * Kl].view(Ml, Kl, 1, 1)
                if ctx.has_bias[l]:
                    grads[2 * l + 1] = flat[offs[i] + Ml * Kl:offs[i] + Ml * Kl + Ml]
        return (dx, None, *grads)


def _mlp64_supported(cin, widths):
    return (os.environ.get('FGNN_MLP64', '1') != '0' and len(widths) == 3 and all(m == 64 for m in widths)
            and bool(_lib.load().fgnn_mlp64_supported(cin, 3, 64)))


class _Mlp64Fn(torch.autograd.Function):
    """The conv stack of a 64-wide MlpBlock_Real (models/layers.py:125-131 with depth 3, 64 hidden / output channels; input 1..32,
    64 or 128 channels: the three MLPs of a 64-feature block) as ONE launch per direction (csrc/mlp64.hip, 16-pixel tiles on
    v_mfma_f32_16x16x4_f32).  Only x is kept for the backward: the hidden activations are recomputed there with the forward's
    fma sequence, the weight gradients are accumulated in the waves' registers (no d(pre-activation) tensors, no second pass)."""

    @staticmethod
    def _args(x, xb, nvalid, packed):
        G, K0, N, _ = x.shape
        P = N * N
        a = _lib.Mlp64Args()
        a.x, a.x_gstride, a.x_ld, a.cin = x.data_ptr(), K0 * P, P, K0
        if xb is not None:
            a.xb, a.xb_gstride, a.xb_ld, a.cb = xb.data_ptr(), xb.shape[1] * P, P, xb.shape[1]
        a.packed = packed.data_ptr()
        a.nvalid = nvalid.data_ptr() if nvalid is not None else None
        a.G, a.N = G, N
        return a

    @staticmethod
    def forward(ctx, x, xb, nvalid, packed, sink_a, sink_b, *wb):
        """xb: None, or a second tensor stacked after x along the channels (Concat's parts: the block's [mult ; in] without the copy).
        packed: None, or the operand record of these parameters (prepack64: one launch for every MLP of a model).
        sink_a / sink_b: None, or the GradSink of x / xb (fan_out): the backward then ADDS its input gradient to the sink's buffer inside
        the kernel and returns nothing for that input -- the readers of one tensor share one gradient buffer, no add kernels."""
        ctx.sinks = (sink_a, sink_b)
        _check(x, 'MlpBlock_Real')
        x = x.contiguous()
        G, Ka, N, _ = x.shape
        P = N * N
        if xb is not None:
            _check(xb, 'MlpBlock_Real')
            xb = xb.contiguous()
            if xb.shape[0] != G or xb.shape[2:] != x.shape[2:]:
                raise RuntimeError('MlpBlock_Real: stacked inputs of different shapes %s / %s' % (tuple(x.shape), tuple(xb.shape)))
        K0 = Ka + (xb.shape[1] if xb is not None else 0)
        ws = [wb[2 * l].contiguous() for l in range(3)]
        bs = [wb[2 * l + 1].contiguous() if wb[2 * l + 1] is not None else None for l in range(3)]
        if ws[0].shape[1] != K0:
            raise RuntimeError('MlpBlock_Real: conv 0 expects %d input channels, gets %d' % (ws[0].shape[1], K0))
        st = _lib.stream_ptr()
        f32 = dict(dtype=torch.float32, device=x.device)
        if packed is None:
            packed = torch.empty(_lib.load().fgnn_mlp64_packed_floats(K0), **f32)      # operand record of both directions
            _lib.call('fgnn_mlp64_pack', _lib.ptr(ws[0]), _lib.ptr(ws[1]), _lib.ptr(ws[2]), *[_lib.ptr(b) if b is not None else None for b in bs],
                      K0, _lib.ptr(packed), st)
        elif packed.numel() != _lib.load().fgnn_mlp64_packed_floats(K0):
            raise RuntimeError('MlpBlock_Real: operand record of %d floats for %d input channels' % (packed.numel(), K0))
        out = torch.empty(G, 64, N, N, **f32)
        a = _Mlp64Fn._args(x, xb, nvalid, packed)
        a.out, a.o_gstride, a.o_ld = out.data_ptr(), 64 * P, P
        _lib.call('fgnn_mlp64_fwd', C.byref(a), st)
        ctx.save_for_backward(x, xb, nvalid, packed)
        ctx.has_bias = [b is not None for b in bs]
        return out

    @staticmethod
    def backward(ctx, dz):
        x, xb, nvalid, packed = ctx.saved_tensors
        dz = dz.contiguous()
        G, Ka, N, _ = x.shape
        Kb = xb.shape[1] if xb is not None else 0
        K0 = Ka + Kb
        P = N * N
        f32 = dict(dtype=torch.float32, device=x.device)
        st = _lib.stream_ptr()
        a = _Mlp64Fn._args(x, xb, nvalid, packed)
        a.dz, a.dz_gstride, a.dz_ld = dz.data_ptr(), 64 * P, P
        dx = dxb = None
        ret_a = ret_b = None
        if ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1]):
            sink_a, sink_b = ctx.sinks

            def target(sink, ch, wanted):
                """(buffer, accumulate?, what autograd gets): the sink's shared buffer, or a fresh tensor"""
                if sink is None or not wanted:
                    t = torch.empty(G, ch, N, N, **f32)
                    return t, 0, (t if wanted else None)
                first = sink.buf is None
                if first:
                    sink.buf = torch.empty(G, ch, N, N, **f32)
                return sink.buf, int(not first), None
            dx, a.accumulate_dx, ret_a = target(sink_a, Ka, ctx.needs_input_grad[0])
            a.dx, a.dx_gstride, a.dx_ld = dx.data_ptr(), Ka * P, P
            if xb is not None:
                dxb, a.accumulate_dxb, ret_b = target(sink_b, Kb, ctx.needs_input_grad[1])
                a.dxb, a.dxb_gstride, a.dxb_ld = dxb.data_ptr(), Kb * P, P
        nwg = _lib.load().fgnn_mlp64_num_workgroups()
        cnt = _lib.load().fgnn_mlp64_param_count(K0)
        wpart = torch.empty(nwg * cnt, **f32)
        flat = torch.empty(cnt, **f32)
        a.wpart = wpart.data_ptr()
        _lib.call('fgnn_mlp64_bwd', C.byref(a), st)
        _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), nwg, cnt, _lib.ptr(flat), st)
        k0p = (K0 + 31) // 32 * 32
        o1 = 64 * k0p + 64
        o2 = o1 + 64 * 64 + 64
        dw0 = flat[:64 * k0p].view(64, k0p)[:, :K0]
        grads = [dw0.reshape(64, K0, 1, 1),
                 flat[64 * k0p:o1] if ctx.has_bias[0] else None,
                 flat[o1:o1 + 4096].view(64, 64, 1, 1), flat[o1 + 4096:o2] if ctx.has_bias[1] else None,
                 flat[o2:o2 + 4096].view(64, 64, 1, 1), flat[o2 + 4096:o2 + 4160] if ctx.has_bias[2] else None]
        return (ret_a, ret_b, None, None, None, None, *grads)


def prepack64(blocks):
    """Operand records (fgnn_mlp64_pack_multi) of every 64-wide MlpBlock_Real in `blocks` in ONE launch per 16 of them; each block's next
    forward call uses its record instead of packing by itself (network.Network.forward calls this at the top of a pass)."""
    todo = [m for m in blocks if not m.fused() and m.convs[0].weight.is_cuda
            and _mlp64_supported(m.convs[0].in_channels, [c.out_channels for c in m.convs])]
    if not todo:
        return
    lib = _lib.load()
    sizes = [lib.fgnn_mlp64_packed_floats(m.convs[0].in_channels) for m in todo]
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=todo[0].convs[0].weight.device)
    off, keep = 0, []
    for i0 in range(0, len(todo), 16):
        chunk = todo[i0:i0 + 16]
        jobs = (_lib.Mlp64PackJob * len(chunk))()
        for jb, m, n in zip(jobs, chunk, sizes[i0:i0 + 16]):
            for l, conv in enumerate(m.convs):
                w = conv.weight if conv.weight.is_contiguous() else conv.weight.contiguous()
                keep.append(w)
                jb.W[l] = w.data_ptr()
                jb.bias[l] = conv.bias.data_ptr() if conv.bias is not None else None
            jb.cin = m.convs[0].in_channels
            m._packed64 = buf[off:off + n]
            jb.packed = m._packed64.data_ptr()
            off += n
        _lib.call('fgnn_mlp64_pack_multi', jobs, len(chunk), _lib.stream_ptr())


class GradSink:
    """One gradient buffer for a tensor that several fused MLPs read (a block's input feeds mlp1, mlp2 and, through Concat, mlp3:
    models/blocks_emb.py:29-36).  fan_out(x) returns x behind an autograd node that owns the sink; every _Mlp64Fn reading it adds
    its input gradient to sink.buf inside its kernel (the first one stores) and hands autograd nothing; the node's backward returns the
    buffer (plus whatever other readers sent the ordinary way)."""

    def __init__(self):
        self.buf = None


class _FanOutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sink):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.sink.buf = ctx.sink.buf, None
        if buf is None:
            return g, None
        return (buf if g is None else buf + g), None


def fan_out(x):
    """x (tensor or MaskedTensor) with a GradSink attached for the fused MLPs that read it (only when it takes part in autograd)."""
    t, _ = _split(x)
    if not (torch.is_tensor(t) and t.requires_grad and torch.is_grad_enabled() and t.is_cuda and t.dtype == torch.float32):
        return x
    sink = GradSink()
    tf = _FanOutFn.apply(t, sink)
    tf._fgnn_sink = sink
    return _wrap(tf, x)


class LazyCat:
    """Concat's output before anyone needs it as ONE tensor: the parts (tensors or MaskedTensors).  The dict-graph executor
    (network.Network.forward) hands it to an MlpBlock_Real whose fused kernels read the parts in place (csrc/mlp64.hip, two slabs);
    every other reader gets torch.cat through materialize()."""

    def __init__(self, parts):
        self.parts = list(parts)
        self._cat = None

    def materialize(self):
        if self._cat is None:
            self._cat = _wrap(torch.cat([_split(x)[0] for x in self.parts], dim=1), self.parts[0])
        return self._cat


# --------------------------------------------------------------------------------------
# GraphNorm / normalize on an arbitrary (G, C, N, N) tensor
# --------------------------------------------------------------------------------------
class _GraphNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nvalid, eps, gn_w, gn_b):
        _check(x, 'GraphNorm')
        x = x.contiguous()
        G, Cc, N, _ = x.shape
        P = N * N
        nrm = torch.empty(G * Cc * 4, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        st = _lib.stream_ptr()
        if _lib.load().fgnn_gn_plane_supported(N):      # one pass, one launch: a (g, c) plane fits a workgroup's registers
            _lib.call('fgnn_gn_plane_fwd', _lib.ptr(x), Cc * P, P, _lib.ptr(gn_w), _lib.ptr(gn_b), _nv(nvalid), G, Cc, N,
                      float(eps), _lib.ptr(y), Cc * P, P, _lib.ptr(nrm), st)
        else:
            _lib.call('fgnn_gn_stats', _lib.ptr(x), Cc * P, P, _lib.ptr(gn_w), _nv(nvalid), G, Cc, N, float(eps),
                      _lib.ptr(nrm), st)
            _lib.call('fgnn_gn_apply', _lib.ptr(x), Cc * P, P, _lib.ptr(nrm), _lib.ptr(gn_b), _nv(nvalid), G, Cc, N,
                      _lib.ptr(y), Cc * P, P, st)
        ctx.save_for_backward(x, nrm, nvalid, gn_w, gn_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, nrm, nvalid, gn_w, gn_b = ctx.saved_tensors
        dy = dy.contiguous()
        G, Cc, N, _ = x.shape
        P = N * N
        f32 = dict(dtype=torch.float32, device=x.device)
        s12 = torch.empty(G * Cc * 2, **f32)
        coef = torch.empty(G * Cc * 4, **f32)
        dgw = torch.empty(Cc, **f32) if gn_w is not None else None
        dgb = torch.empty(Cc, **f32) if gn_b is not None else None
        dx = torch.empty_like(x)
        st = _lib.stream_ptr()
        gs = Cc * P
        if _lib.load().fgnn_gn_plane_supported(N):
            _lib.call('fgnn_gn_plane_bwd', _lib.ptr(dy), gs, P, _lib.ptr(x), gs, P, _lib.ptr(nrm), _nv(nvalid), G, Cc, N,
                      _lib.ptr(dx), gs, P, _lib.ptr(s12), _lib.ptr(dgw), _lib.ptr(dgb), st)
        else:
            _lib.call('fgnn_gn_bwd_stats', _lib.ptr(dy), gs, P, _lib.ptr(x), gs, P, _lib.ptr(nrm), _nv(nvalid), G, Cc, N,
                      _lib.ptr(s12), st)
            _lib.call('fgnn_gn_bwd_coef', _lib.ptr(s12), _lib.ptr(nrm), _nv(nvalid), G, Cc, N, _lib.ptr(coef),
                      _lib.ptr(dgw), _lib.ptr(dgb), st)
            _lib.call('fgnn_gn_bwd_apply', _lib.ptr(dy), gs, P, _lib.ptr(x), gs, P, _lib.ptr(coef), _nv(nvalid), G, Cc, N,
                      _lib.ptr(dx), gs, P, st)
        return (dx, None, None, dgw.view(1, Cc, 1, 1) if dgw is not None else None,
                dgb.view(1, Cc, 1, 1) if dgb is not None else None)


def normalize(b, constant_n_vertices=True, eps=1e-05):
    """(b - mean) / (2 sqrt(n (var + eps))) per (graph, channel)  (models/layers.py:71-80)."""
    x, nvalid = _split(b)
    if not constant_n_vertices and nvalid is None:
        raise RuntimeError('normalize(constant_n_vertices=False) expects a MaskedTensor')
    return _wrap(_GraphNormFn.apply(x, nvalid, eps, None, None), b)


class GraphNorm(nn.Module):
    def __init__(self, features, constant_n_vertices=True, elementwise_affine=True, eps=1e-05, device=None, dtype=None):
        super().__init__()
        factory_kwargs = {'device': device, 'dtype': dtype}
        self.constant_n_vertices = constant_n_vertices
        self.eps = eps
        self.elementwise_affine = elementwise_affine
        self.features = (1, features, 1, 1)
        if elementwise_affine:
            self.weight = Parameter(torch.ones(self.features, **factory_kwargs))
            self.bias = Parameter(torch.zeros(self.features, **factory_kwargs))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)

    def reset_parameters(self):
        if self.elementwise_affine:
            nn.init.ones_(self.weight)
            nn.init.zeros_(self.bias)

    def forward(self, b):
        x, nvalid = _split(b)
        return _wrap(_GraphNormFn.apply(x, nvalid, self.eps, self.weight, self.bias), b)


def _init_weights(layer):
    nn.init.xavier_uniform_(layer.weight)
    if layer.bias is not None:
        nn.init.zeros_(layer.bias)


class MlpBlock_Real(nn.Module):
    """depth_of_mlp 1x1 convs with ReLU after all but the last, then GraphNorm."""

    def __init__(self, in_features, out_features, depth_of_mlp, activation_fn=F.relu, constant_n_vertices=True):
        super().__init__()
        if activation_fn is not F.relu:
            raise RuntimeError('MlpBlock_Real: the fused HIP kernels implement ReLU only')
        self.activation = activation_fn
        self.depth_mlp = depth_of_mlp
        self.cst_vertices = constant_n_vertices
        self.convs = nn.ModuleList()
        for _ in range(depth_of_mlp):
            self.convs.append(nn.Conv2d(in_features, out_features, kernel_size=1, padding=0, bias=True))
            _init_weights(self.convs[-1])
            in_features = out_features
        self.gn = GraphNorm(out_features, constant_n_vertices=constant_n_vertices)

    def fused(self):
        """True when the fused 32-wide kernels (mlp_fwd.hip / mlp_bwd.hip) are built for this block's widths."""
        return (all(c.out_channels == FGNN_H and c.bias is not None for c in self.convs)
                and self.convs[0].in_channels in FUSED_INPUT_WIDTHS and 1 <= len(self.convs) <= _lib.FGNN_MAX_DEPTH
                and self.gn.features[1] == FGNN_H)

    def _take_packed64(self):
        """the operand record prepack64 left for this forward call (used once), or None"""
        p, self._packed64 = getattr(self, '_packed64', None), None
        return p

    def takes_parts(self, parts):
        """True when forward() reads Concat's parts in place (two stacked slabs of the fused 64-wide kernels: 64 channels, then <= 64)."""
        if self.fused() or len(parts) != 2:
            return False
        ta, tb = _split(parts[0])[0], _split(parts[1])[0]
        return (_mlp64_supported(self.convs[0].in_channels, [c.out_channels for c in self.convs]) and ta.dim() == 4 and tb.dim() == 4
                and ta.shape[1] == 64 and 1 <= tb.shape[1] <= 64 and ta.shape[1] + tb.shape[1] == self.convs[0].in_channels
                and ta.is_cuda and tb.is_cuda and ta.dtype == torch.float32 and tb.dtype == torch.float32)

    def forward(self, inputs):
        if isinstance(inputs, LazyCat):
            if not self.takes_parts(inputs.parts):
                return self.forward(inputs.materialize())
            (xa, nvalid), (xb, _) = _split(inputs.parts[0]), _split(inputs.parts[1])
            wb = []
            for conv in self.convs:
                wb += [conv.weight, conv.bias]
            y = _Mlp64Fn.apply(xa, xb, nvalid, self._take_packed64(), getattr(xa, '_fgnn_sink', None), getattr(xb, '_fgnn_sink', None), *wb)
            y = _GraphNormFn.apply(y, nvalid, self.gn.eps, self.gn.weight, self.gn.bias)
            return _wrap(y, inputs.parts[0])
        x, nvalid = _split(inputs)
        if self.fused():
            wb = []
            for conv in self.convs:
                wb += [conv.weight, conv.bias]
            y = _MlpGnFn.apply(x, nvalid, self.gn.eps, self.gn.weight, self.gn.bias, *wb)
        else:       # any other widths (csrc/conv.hip), then GraphNorm on the (G, C, N, N) tensor
            if _mlp64_supported(self.convs[0].in_channels, [c.out_channels for c in self.convs]):
                wb = []
                for conv in self.convs:
                    wb += [conv.weight, conv.bias]
                # 64-wide stacks: fused, hidden activations recomputed in the backward
                y = _Mlp64Fn.apply(x, None, nvalid, self._take_packed64(), getattr(x, '_fgnn_sink', None), None, *wb)
            elif _chain_supported(self.convs[0].in_channels, [c.out_channels for c in self.convs]):
                wb = []
                for conv in self.convs:
                    wb += [conv.weight, conv.bias]
                y = _ConvChainFn.apply(x, nvalid, *wb)         # the whole conv stack in one launch per direction
            else:
                y = x
                last = len(self.convs) - 1
                for l, conv in enumerate(self.convs):
                    y = _ConvFn.apply(y, nvalid, conv.weight, conv.bias, l < last)
            y = _GraphNormFn.apply(y, nvalid, self.gn.eps, self.gn.weight, self.gn.bias)
        return _wrap(y, inputs)


# --------------------------------------------------------------------------------------
# Matmul / Concat / ColumnMaxPooling / Identity
# --------------------------------------------------------------------------------------
class _MatmulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, nvalid):
        _check(a, 'Matmul')
        _check(b, 'Matmul')
        a, b = a.contiguous(), b.contiguous()
        G, Cc, N, _ = a.shape
        P = N * N
        out = torch.empty_like(a)
        sa, sb = _lib.make_slab(a, Cc * P, P, Cc), _lib.make_slab(b, Cc * P, P, Cc)
        _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), _nv(nvalid), G, N, _lib.ptr(out), Cc * P, P,
                  _lib.stream_ptr())
        ctx.save_for_backward(a, b, nvalid)
        return out

    @staticmethod
    def backward(ctx, dm):
        a, b, nvalid = ctx.saved_tensors
        dm = dm.contiguous()
        G, Cc, N, _ = a.shape
        P = N * N
        da, db = torch.empty_like(a), torch.empty_like(b)
        sa, sb = _lib.make_slab(a, Cc * P, P, Cc), _lib.make_slab(b, Cc * P, P, Cc)
        _lib.call('fgnn_chan_matmul_bwd', C.byref(sa), C.byref(sb), _lib.ptr(dm), Cc * P, P, _nv(nvalid), G, N,
                  _lib.ptr(da), _lib.ptr(db), Cc * P, P, None, None, _lib.stream_ptr())
        return da, db, None


class Matmul(nn.Module):
    def forward(self, xs1, xs2):
        a, nvalid = _split(xs1)
        b, _ = _split(xs2)
        return _wrap(_MatmulFn.apply(a, b, nvalid), xs1)


class Concat(nn.Module):
    """torch.cat on channels (padding stays zero, so no re-mask is needed)."""

    def forward(self, *xs):
        ts = [_split(x)[0] for x in xs]
        return _wrap(torch.cat(ts, dim=1), xs[0])


class _ColMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nvalid):
        _check(x, 'ColumnMaxPooling')
        x = x.contiguous()
        G, Cc, N, _ = x.shape
        P = N * N
        e = torch.empty(G, Cc, N, dtype=torch.float32, device=x.device)
        idx = torch.empty(G, Cc, N, dtype=torch.int32, device=x.device)
        s = _lib.make_slab(x, Cc * P, P, Cc)
        _lib.call('fgnn_colmax_fwd', C.byref(s), _nv(nvalid), G, N, _lib.ptr(e), _lib.ptr(idx), _lib.stream_ptr())
        ctx.save_for_backward(idx, nvalid)
        ctx.n = N
        return e

    @staticmethod
    def backward(ctx, de):
        idx, nvalid = ctx.saved_tensors
        de = de.contiguous()
        G, Cc, N = de.shape
        P = N * N
        dx = torch.empty(G, Cc, N, N, dtype=torch.float32, device=de.device)
        _lib.call('fgnn_colmax_bwd', _lib.ptr(de), _lib.ptr(idx), _nv(nvalid), G, Cc, N, _lib.ptr(dx), Cc * P, P,
                  None, None, _lib.stream_ptr())
        return dx, None


class ColumnMaxPooling(nn.Module):
    """(bs, features, n, n) -> (bs, features, n): max over the last index."""

    def forward(self, x):
        t, nvalid = _split(x)
        return _wrap(_ColMaxFn.apply(t, nvalid), x, masked_dims=(2,))


class Identity(namedtuple('Identity', [])):
    def __call__(self, x):
        return x
