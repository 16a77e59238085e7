"""bf16 execution of the 2-FGNN hot path on one MI355X (BASELINE config 4: N = 200 dense pairs in 16-bit).

Same launch structure as ``engine.FgnnEngine`` -- what ``Siamese_Node_Exp.forward`` + ``triplet_loss`` + autograd do
in the reference (models/trainers.py:60-68, models/blocks_emb.py:16-43, toolbox/losses.py:20-34), which trains
under 16-bit AMP (commander_explore.py:120-122, ``Network.half`` models/utils.py:71-74) -- with every activation
and gradient slab STORED as bf16 (half the HBM bytes of the regime SURVEY.md 8d calls purely HBM-bound), every
channel contraction and per-channel N x N product on ``v_mfma_f32_32x32x16_bf16``, and fp32 accumulation,
biases, GraphNorm statistics, embeddings, scores, loss and parameter gradients.  Parameters and gradients are the
same flat fp32 buffers as in the fp32 engine (one all-reduce per step).

Rounding points are spelled out in ``oracle/fgnn_oracle_bf16.py`` (test infrastructure; never imported here).
"""
import ctypes as C
import os

import torch

from . import _lib
from .engine import EPS, ParamLayout, _round_up  # noqa: F401


class FgnnEngineBF16:
    """Workspace + launch sequence for a fixed (G, N) problem on the current device, bf16 storage."""
    SKIP_PADDING_TILES = True     # ragged engines: fgnn_ragged_tile_ranges16 + tile skipping in fgnn_mlp_fwd16 / fgnn_mlp_bwd16
    PAIR_BWD = True               # mlp1 + mlp2 backward of a block as one launch (fgnn_mlp_bwd16_pair), constant-size batches
    BLOCK1 = os.environ.get('FGNN_BLOCK1', 'generic')      # 'structured': csrc/block1_struct.hip for bit-packed inputs
    PACK_IN_STRUCT = os.environ.get('FGNN_PACK_IN_STRUCT', '1') != '0'      # as FgnnEngine.PACK_IN_STRUCT

    def __init__(self, layout, G, N, device, ragged=False, block1=None):
        lib = _lib.load()
        block1 = self.BLOCK1 if block1 is None else block1
        if block1 not in ('generic', 'structured'):
            raise ValueError("block1 must be 'generic' or 'structured' (got %r)" % (block1,))
        # the structured block 1 applies to bit-packed inputs (embed(bits=...)), constant-size or ragged; dense inputs run generic
        self.struct1 = block1 == 'structured' and bool(lib.fgnn_block1_struct_supported(N, layout.depth, layout.c0))
        self._struct = None
        self.xbits = None
        if layout.depth != 3:
            raise RuntimeError('the bf16 kernels are built for depth_of_mlp = 3 (got %d)' % layout.depth)
        if layout.c0 != 2:
            raise RuntimeError('the bf16 kernels are built for original_features_num = 2 (got %d)' % layout.c0)
        if N > 256:
            raise RuntimeError('the bf16 per-channel matmul handles N <= 256 (got %d)' % N)
        self.layout = layout
        self.G, self.N = G, N
        self.B = G // 2
        self.ldr = _round_up(N, 8)                       # row pitch of an N x N matrix inside a channel
        self.ldp = _round_up(N * self.ldr, 64)           # channel stride
        self.tpg = lib.fgnn_tiles_per_graph16(N, self.ldr)
        self.device = device
        K = layout.num_blocks
        f32 = dict(dtype=torch.float32, device=device)
        bf = dict(dtype=torch.bfloat16, device=device)
        act = lambda: torch.empty(G * 32 * self.ldp, **bf)
        self.x16 = torch.empty(G * 2 * self.ldp, **bf)
        self.z = {(k, j): act() for k in range(1, K + 1) for j in (1, 2, 3)}
        self.mult = {k: act() for k in range(1, K + 1)}
        self.nrm = {(k, j): torch.empty(G * 32 * 4, **f32) for k in range(1, K + 1) for j in (1, 2, 3)}
        self.part = [torch.empty(G * self.tpg * 32 * 2, **f32) for _ in range(2)]
        self.cnt = torch.empty(G * self.tpg, **f32)
        self.E = torch.empty(G, 32, N, **f32)
        self.idx = torch.empty(G, 32, N, dtype=torch.int32, device=device)
        self.scores = torch.empty(self.B, N, N, **f32)
        self.lse = torch.empty(self.B, N, **f32)
        self.score_blocks = lib.fgnn_score_row_blocks(self.B, N)      # row blocks per pair of the scoring kernel
        self.pair_loss = torch.empty(self.B * self.score_blocks, **f32)
        self.loss = torch.empty(1, **f32)
        self.nvalid = torch.empty(G, dtype=torch.int32, device=device) if ragged else None
        self._nvalid_own = self.nvalid      # the engine's own buffer; an int32 device tensor handed in is used in place (no copy launch)
        # ragged batches: work-balanced tile ranges of the MLP kernels (padding-only tiles are stepped over)
        self.ranges = (torch.empty(_lib.FGNN_RANGE_WG + 1, dtype=torch.int32, device=device)
                       if ragged and self.SKIP_PADDING_TILES else None)
        self._bwd = None
        self._packs = {}
        for k in range(1, K + 1):
            cin = layout.c0 if k == 1 else 32
            self._packs[('f', k, 12)] = (0, cin, 0, 2, torch.empty(lib.fgnn_pack16_floats(0, cin, 0, 3, 2), **f32))
            self._packs[('f', k, 3)] = (0, 32, cin, 1, torch.empty(lib.fgnn_pack16_floats(0, 32, cin, 3, 1), **f32))
            for j in (1, 2):
                self._packs[('b', k, j)] = (1, cin, 0, 1, torch.empty(lib.fgnn_pack16_floats(1, cin, 0, 3, 1), **f32))
            self._packs[('b', k, 3)] = (1, 32, cin, 1, torch.empty(lib.fgnn_pack16_floats(1, 32, cin, 3, 1), **f32))

    # ------------------------------------------------------------------ helpers
    def _nv(self):
        return _lib.ptr(self.nvalid) if self.nvalid is not None else None

    def _w(self, params, off):
        return params.data_ptr() + 4 * off

    def _slab_in(self, k, params):
        if k == 1:
            return _lib.make_slab16(self.x16, 2 * self.ldp, self.ldp, 2)
        rec = self.layout.mlp[(k - 1, 3)]
        return _lib.make_slab16(self.z[(k - 1, 3)], 32 * self.ldp, self.ldp, 32, nrm=self.nrm[(k - 1, 3)],
                                beta=self._w(params, rec['gn_b']))

    def _slab_z(self, k, j, params):
        rec = self.layout.mlp[(k, j)]
        return _lib.make_slab16(self.z[(k, j)], 32 * self.ldp, self.ldp, 32, nrm=self.nrm[(k, j)],
                                beta=self._w(params, rec['gn_b']))

    def _slab_raw(self, t):
        return _lib.make_slab16(t, 32 * self.ldp, self.ldp, 32)

    def _pack_jobs(self, params, chunk):
        L = self.layout
        jobs = (_lib.PackJob * len(chunk))()
        for i, ((kind, k, which), (knd, ca, cb, nmlp, buf)) in enumerate(chunk):
            jobs[i].kind, jobs[i].ca, jobs[i].cb, jobs[i].depth, jobs[i].nmlp = knd, ca, cb, L.depth, nmlp
            js = (1, 2) if which == 12 else (which,)
            for m, j in enumerate(js):
                rec = L.mlp[(k, j)]
                for l in range(L.depth):
                    jobs[i].W[m][l] = self._w(params, rec['w'][l])
                    jobs[i].bias[m][l] = self._w(params, rec['b'][l])
            jobs[i].out = buf.data_ptr()
        return jobs

    def pack_operands(self, params):
        items = list(self._packs.items())
        for lo in range(0, len(items), _lib.MAX_PACK_JOBS):
            chunk = items[lo:lo + _lib.MAX_PACK_JOBS]
            _lib.call('fgnn_pack16_operands', self._pack_jobs(params, chunk), len(chunk), _lib.stream_ptr())

    def _mlp_fwd(self, params, k, js, a, b, finalize=True):
        """finalize=False: leave the tile statistics un-finalized (the matmul that consumes the two outputs finalizes them in
        its own prologue, fgnn_chan_matmul_fwd16_fin)."""
        L = self.layout
        args = _lib.MlpFwd16Args()
        args.G, args.N, args.ldr, args.depth, args.nmlp = self.G, self.N, self.ldr, L.depth, len(js)
        args.nvalid = self.nvalid.data_ptr() if self.nvalid is not None else None
        args.a = a
        if b is not None:
            args.b = b
        for m, j in enumerate(js):
            args.z[m] = self.z[(k, j)].data_ptr()
            args.part[m] = self.part[m].data_ptr()
        args.ldz = self.ldp
        args.cnt = self.cnt.data_ptr()
        args.packed = self._packs[('f', k, 12 if len(js) == 2 else 3)][4].data_ptr()
        if self.ranges is not None:
            args.ranges = self.ranges.data_ptr()
        st = _lib.stream_ptr()
        _lib.call('fgnn_mlp_fwd16', C.byref(args), st,
                  tag='mlp_fwd16[cin=%d,nmlp=%d]' % (a.C + (b.C if b is not None else 0), len(js)))
        if getattr(self, 'decisions', None) is not None:
            # test-only (export_decisions): the decision-exporting twin of the launch above -- same tile code, same outputs written once
            # more -- leaves one bit per hidden pre-activation of these MLPs
            bufs = [torch.zeros(self.G * (L.depth - 1) * 32 * self.tpg * 2, dtype=torch.int32, device=self.device) for _ in js]
            _lib.call('fgnn_debug_mlp_fwd16_masks', C.byref(args), _lib.ptr(bufs[0]), _lib.ptr(bufs[1]) if len(js) == 2 else None, st)
            for j, buf in zip(js, bufs):
                self.decisions[(k, j)] = buf
        if not finalize:
            return
        if len(js) == 2:
            r0, r1 = L.mlp[(k, js[0])], L.mlp[(k, js[1])]
            _lib.call('fgnn_gn_finalize2_tpg', _lib.ptr(self.part[0]), _lib.ptr(self.part[1]), _lib.ptr(self.cnt),
                      C.c_void_p(self._w(params, r0['gn_w'])), C.c_void_p(self._w(params, r1['gn_w'])), self._nv(),
                      self.G, 32, self.N, self.tpg, EPS, _lib.ptr(self.nrm[(k, js[0])]), _lib.ptr(self.nrm[(k, js[1])]), st)
        else:
            rec = L.mlp[(k, js[0])]
            _lib.call('fgnn_gn_finalize_tpg', _lib.ptr(self.part[0]), _lib.ptr(self.cnt),
                      C.c_void_p(self._w(params, rec['gn_w'])), self._nv(), self.G, 32, self.N, self.tpg, EPS,
                      _lib.ptr(self.nrm[(k, js[0])]), st)

    # ------------------------------------------------------------------ forward
    def embed(self, params, x, nvalid=None, bits=None):
        """x: (G, 2, N, N) contiguous fp32 device tensor (0/1 adjacency + degrees: exact in bf16 for N <= 256) -- or bits:
        (G, N, ceil(N/32)) int32 words of the bit-packed adjacency (the input form of FgnnEngine.embed(bits=...)); with
        block1='structured' block 1 then runs on the class tables (csrc/block1_struct.hip).
        LIFETIME: as FgnnEngine.embed -- bits and an int32 device nvalid of G entries are read in place by this forward and by the
        backward after it; leave them untouched until backward() has been issued."""
        L = self.layout
        if (nvalid is None) != (self.nvalid is None):
            raise RuntimeError('FgnnEngineBF16: ragged flag and nvalid argument disagree')
        if bits is not None:
            words = (self.N + 31) // 32
            if x is not None:
                raise RuntimeError('FgnnEngineBF16.embed: pass x or bits, not both')
            if tuple(bits.shape) != (self.G, self.N, words) or bits.dtype not in (torch.int32, torch.uint32) \
                    or not bits.is_contiguous() or bits.device.type != 'cuda':
                raise RuntimeError('FgnnEngineBF16.embed: expected contiguous int32 device bits %s, got %s %s'
                                   % ((self.G, self.N, words), tuple(bits.shape), bits.dtype))
            if not self.struct1:
                raise RuntimeError("FgnnEngineBF16.embed: bits= needs block1='structured' (N <= 256, depth 3, 2 input channels)")
        elif x.shape != (self.G, L.c0, self.N, self.N) or not x.is_contiguous() or x.dtype != torch.float32:
            raise RuntimeError('FgnnEngineBF16.embed: expected contiguous fp32 %s, got %s %s'
                               % ((self.G, L.c0, self.N, self.N), tuple(x.shape), x.dtype))
        self.xbits = bits
        if nvalid is not None:
            if nvalid.dtype == torch.int32 and nvalid.is_cuda and nvalid.is_contiguous() and nvalid.numel() == self.G:
                self.nvalid = nvalid            # read in place by every kernel of the step (a copy node costs 4.6 + 8.6 us of gap in a replayed graph)
            else:
                self._nvalid_own.copy_(nvalid.to(torch.int32))
                self.nvalid = self._nvalid_own
        st = _lib.stream_ptr()
        if self.ranges is not None:
            _lib.call('fgnn_ragged_tile_ranges16', _lib.ptr(self.nvalid), self.G, self.N, self.ldr, _lib.ptr(self.ranges), st)
        if bits is None:
            _lib.call('fgnn_to_bf16', _lib.ptr(x), self._nv(), self.G, 2, self.N, self.ldr, _lib.ptr(self.x16),
                      2 * self.ldp, self.ldp, st)
        # the structured block 1's first launch carries the packing as extra workgroups (fgnn_block1_struct_fwd16_pack: one launch less)
        pack_in_struct = bits is not None and self.PACK_IN_STRUCT and len(self._packs) <= _lib.MAX_PACK_JOBS
        if not pack_in_struct:
            self.pack_operands(params)
        gs = 32 * self.ldp
        for k in range(1, L.num_blocks + 1):
            sin = self._slab_in(k, params)
            if k == 1 and bits is not None:
                self._struct_fwd(params, with_pack=pack_in_struct)      # mlp1 + mlp2 + mult of block 1 from the class tables; also writes x16
                self._mlp_fwd(params, k, (3,), self._slab_raw(self.mult[k]), sin)
                continue
            # mlp1 / mlp2's GraphNorm records are finalized by the matmul that consumes them (one launch less per block)
            self._mlp_fwd(params, k, (1, 2), sin, None, finalize=False)
            ya, yb = self._slab_z(k, 1, params), self._slab_z(k, 2, params)
            r1, r2 = L.mlp[(k, 1)], L.mlp[(k, 2)]
            _lib.call('fgnn_chan_matmul_fwd16_fin', C.byref(ya), C.byref(yb), _lib.ptr(self.part[0]), _lib.ptr(self.part[1]),
                      _lib.ptr(self.cnt), C.c_void_p(self._w(params, r1['gn_w'])), C.c_void_p(self._w(params, r2['gn_w'])), EPS,
                      self.tpg, self._nv(), self.G, self.N, self.ldr, _lib.ptr(self.mult[k]), gs, self.ldp, st,
                      tag='fgnn_chan_matmul_fwd16')
            self._mlp_fwd(params, k, (3,), self._slab_raw(self.mult[k]), sin)
        out = self._slab_z(L.num_blocks, 3, params)
        _lib.call('fgnn_colmax_fwd16', C.byref(out), self._nv(), self.G, self.N, self.ldr, _lib.ptr(self.E),
                  _lib.ptr(self.idx), st)
        return self.E

    # ------------------------------------------------------------------ block 1 on its structured input (csrc/block1_struct.hip)
    def _struct_ws(self):
        if self._struct is None:
            lib = _lib.load()
            f32 = dict(dtype=torch.float32, device=self.device)
            self._struct = {'tab': torch.empty(lib.fgnn_block1_struct_table_floats(self.N), **f32),
                            'ws': torch.empty(lib.fgnn_block1_struct_ws_floats(self.G, self.N), **f32)}
        return self._struct

    def _w3(self, params, j):
        rec = self.layout.mlp[(1, j)]
        return ((C.c_void_p * 3)(*[self._w(params, o) for o in rec['w']]), (C.c_void_p * 3)(*[self._w(params, o) for o in rec['b']]))

    def _struct_fwd(self, params, with_pack=False):
        S = self._struct_ws()
        st = _lib.stream_ptr()
        (w1, b1), (w2, b2) = self._w3(params, 1), self._w3(params, 2)
        r1, r2 = self.layout.mlp[(1, 1)], self.layout.mlp[(1, 2)]
        args = [_lib.ptr(self.xbits), self._nv(), self.G, self.N, self.ldr, _lib.ptr(S['tab']),
                C.c_void_p(self._w(params, r1['gn_w'])), C.c_void_p(self._w(params, r1['gn_b'])),
                C.c_void_p(self._w(params, r2['gn_w'])), C.c_void_p(self._w(params, r2['gn_b'])), EPS,
                _lib.ptr(self.nrm[(1, 1)]), _lib.ptr(self.nrm[(1, 2)]), _lib.ptr(self.mult[1]), 32 * self.ldp, self.ldp,
                _lib.ptr(self.x16), _lib.ptr(S['ws']), w1, b1, w2, b2]      # (the class tables are built by the same launch)
        if with_pack:       # ... and the operand images of the step's MLP launches (pack_operands, without its launch)
            items = list(self._packs.items())
            _lib.call('fgnn_block1_struct_fwd16_pack', *args, self._pack_jobs(params, items), len(items), st)
        else:
            _lib.call('fgnn_block1_struct_fwd16', *args, st)

    def _struct_bwd(self, params):
        S, W = self._struct_ws(), self._bwd
        W['struct_rows'] = int(_lib.load().fgnn_block1_struct_rows(self.G, self.N))     # the partial rows this backward writes (read by grad_finalize)
        (w1, _), (w2, _) = self._w3(params, 1), self._w3(params, 2)
        r1, r2 = self.layout.mlp[(1, 1)], self.layout.mlp[(1, 2)]
        _lib.call('fgnn_block1_struct_bwd16', _lib.ptr(self.xbits), self._nv(), self.G, self.N, self.ldr, _lib.ptr(S['tab']), w1, w2,
                  _lib.ptr(self.nrm[(1, 1)]), _lib.ptr(self.nrm[(1, 2)]),
                  C.c_void_p(self._w(params, r1['gn_b'])), C.c_void_p(self._w(params, r2['gn_b'])),
                  _lib.ptr(W['dmult']), 32 * self.ldp, self.ldp, _lib.ptr(S['ws']),
                  _lib.ptr(W['wpart'][(1, 1)]), _lib.ptr(W['wpart'][(1, 2)]), _lib.ptr(W['s12'][(1, 1)]), _lib.ptr(W['s12'][(1, 2)]),
                  _lib.stream_ptr())

    def forward(self, params, x, nvalid=None, total_nodes=None, defer_loss=False, loss_out=None, bits=None):
        self.embed(params, x, nvalid, bits=bits)
        B, N = self.B, self.N
        st = _lib.stream_ptr()
        e1, e2 = self.E[:B], self.E[B:]
        _lib.call('fgnn_score_ce_fwd_blocks', _lib.ptr(e1), _lib.ptr(e2), self._nv(), B, 32, N, self.score_blocks,
                  _lib.ptr(self.scores), _lib.ptr(self.lse), _lib.ptr(self.pair_loss), st)
        if total_nodes is None:
            total_nodes = B * N if nvalid is None else int(nvalid[:B].sum().item())
        self.total_nodes = float(total_nodes)
        self._loss_pending = bool(defer_loss)
        self._loss_target = self.loss if loss_out is None else loss_out     # 1-element fp32 device tensor
        if not defer_loss:
            _lib.call('fgnn_sum_scale', _lib.ptr(self.pair_loss), B * self.score_blocks, 1, 1.0 / self.total_nodes,
                      _lib.ptr(self._loss_target), st)
        return self.scores, self._loss_target

    # ------------------------------------------------------------------ backward
    def _alloc_bwd(self):
        if self._bwd is not None:
            return self._bwd
        f32 = dict(dtype=torch.float32, device=self.device)
        bf = dict(dtype=torch.bfloat16, device=self.device)
        act = lambda: torch.empty(self.G * 32 * self.ldp, **bf)
        nwg = _lib.load().fgnn_mlp_bwd_num_workgroups()
        L = self.layout
        keys = [(k, j) for k in range(1, L.num_blocks + 1) for j in (1, 2, 3)]
        self._bwd = {
            'dE': torch.empty(self.G, 32, self.N, **f32),
            'dy': [act(), act()],
            'dmult': act(), 'dy1': act(), 'dy2': act(),
            's12': {kj: torch.empty(self.G * 32 * 2, **f32) for kj in keys},
            'wpart': {kj: torch.empty(nwg * L.mlp[kj]['count'], **f32) for kj in keys},
            's12part': torch.empty(self.G * self.tpg * 32 * 2, **f32),
            'coef': [torch.empty(self.G * 32 * 4, **f32) for _ in range(3)],
            'nwg': nwg,
            'gscale': torch.empty(1, **f32),
        }
        return self._bwd

    def _mlp_bwd(self, params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit=False):
        args = self._mlp_bwd_args(params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit)
        _lib.call('fgnn_mlp_bwd16', C.byref(args), _lib.stream_ptr(),
                  tag='mlp_bwd16[cin=%d,dx=%d]' % (a.C + (b.C if b is not None else 0),
                                                  (a.C if dxa is not None else 0) + (b.C if (b is not None and dxb is not None) else 0)))

    def _mlp_bwd_pair(self, params, k, sin, din, emit):
        """mlp1 + mlp2 of block k in ONE launch (csrc/mlp_bwd16_pair.hip): the stored d_in is bit-identical to the two
        read-modify-write launches it replaces."""
        W = self._bwd
        a1 = self._mlp_bwd_args(params, k, 1, sin, None, W['dy1'], W['coef'][0], None, None, False, False, False)
        a2 = self._mlp_bwd_args(params, k, 2, sin, None, W['dy2'], W['coef'][1], din, None, True, False, emit)
        _lib.call('fgnn_mlp_bwd16_pair', C.byref(a1), C.byref(a2), _lib.stream_ptr(),
                  tag='mlp_bwd16_pair[cin=%d,dx=%d]' % (sin.C, sin.C if din is not None else 0))

    def _mlp_bwd_args(self, params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit):
        L = self.layout
        W = self._bwd
        gs = 32 * self.ldp
        args = _lib.MlpBwd16Args()
        args.G, args.N, args.ldr, args.depth = self.G, self.N, self.ldr, L.depth
        args.nvalid = self.nvalid.data_ptr() if self.nvalid is not None else None
        args.a = a
        if b is not None:
            args.b = b
        args.dy, args.dgstride, args.ldd = dy.data_ptr(), gs, self.ldp
        args.z, args.zgstride, args.ldz = self.z[(k, j)].data_ptr(), gs, self.ldp
        args.coef = coef.data_ptr()
        if dxa is not None:
            args.dxa, args.dxa_gstride, args.dxa_ld = dxa.data_ptr(), gs, self.ldp
        if dxb is not None:
            args.dxb, args.dxb_gstride, args.dxb_ld = dxb.data_ptr(), gs, self.ldp
        args.accumulate_a, args.accumulate_b = int(acc_a), int(acc_b)
        args.wpart = W['wpart'][(k, j)].data_ptr()
        args.packed = self._packs[('b', k, j)][4].data_ptr()
        if self.ranges is not None:
            args.ranges = self.ranges.data_ptr()
        if emit:
            args.s12part = W['s12part'].data_ptr()
        return args

    def backward(self, params, grads, grad_scale=1.0, hook=None, gscale_dev=None):
        """gscale_dev: a 1-element fp32 DEVICE tensor holding grad_scale / total_nodes (replaces both), as in FgnnEngine.backward:
        the normaliser of a ragged batch never visits the host and a captured step survives another node count."""
        W = self._alloc_bwd()
        B, N = self.B, self.N
        st = _lib.stream_ptr()
        gs_t = W['gscale']
        if gscale_dev is not None:
            gs_t = gscale_dev                  # read in place (a 1-element fp32 device tensor; no copy launch)
        else:
            gs = grad_scale / self.total_nodes
            if W.get('gscale_value') != gs:
                W['gscale'].fill_(gs)
                W['gscale_value'] = gs
        e1, e2 = self.E[:B], self.E[B:]
        _lib.call('fgnn_score_ce_bwd', _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(self.scores), _lib.ptr(self.lse),
                  self._nv(), _lib.ptr(gs_t), B, 32, N, _lib.ptr(W['dE'][:B]), _lib.ptr(W['dE'][B:]), st)
        return self.backward_from_dE(params, grads, W['dE'], hook=hook)

    def backward_from_dE(self, params, grads, dE, hook=None):
        """hook(stage, k): optional inspection callback, called after every kernel of the block backward ('colmax_bwd',
        'mlp3_bwd', 'matmul_bwd', 'mlp1_bwd', 'mlp2_bwd'); the kernel-level parity tests use it to compare each kernel's
        output slabs element by element and to substitute the oracle's values for the next kernel's inputs."""
        if hook is None:
            hook = lambda stage, k: None
        # (a hook with the attribute per_mlp = True asks for the two single-MLP launches, to look at mlp1's output alone)
        L = self.layout
        W = self._alloc_bwd()
        st = _lib.stream_ptr()
        gs = 32 * self.ldp
        K = L.num_blocks

        # the dz coefficients of an MLP are written by the kernel that forms its s12 sums (pooling backward for the last MLP,
        # the matmul backward for mlp1 / mlp2): no separate fgnn_gn_bwd_coef launches
        dy = W['dy'][0]
        out = self._slab_z(K, 3, params)
        _lib.call('fgnn_colmax_bwd16_coef', _lib.ptr(dE), _lib.ptr(self.idx), self._nv(), self.G, 32, self.N, self.ldr,
                  _lib.ptr(dy), gs, self.ldp, C.byref(out), _lib.ptr(W['s12'][(K, 3)]), _lib.ptr(W['coef'][2]), st,
                  tag='fgnn_colmax_bwd16')
        hook('colmax_bwd', K)
        coef3 = W['coef'][2]
        for k in range(K, 0, -1):
            sin = self._slab_in(k, params)
            first = (k == 1)
            din = None if first else W['dy'][(K - k + 1) % 2]
            # mlp3's backward also emits the per-tile trace term <dmult, mult>; the matmul backward derives the S2 sums of both
            # its outputs from it instead of re-reading the two raw operand slabs
            self._mlp_bwd(params, k, 3, self._slab_raw(self.mult[k]), sin, dy, coef3, W['dmult'], din, False, False, emit=True)
            hook('mlp3_bwd', k)
            if first and self.xbits is not None:
                self._struct_bwd(params)
                break
            if first:
                W['struct_rows'] = 0                    # the generic kernels fill every partial row of block 1
            ya, yb = self._slab_z(k, 1, params), self._slab_z(k, 2, params)
            _lib.call('fgnn_chan_matmul_bwd16_tc', C.byref(ya), C.byref(yb), _lib.ptr(W['dmult']), gs, self.ldp,
                      _lib.ptr(W['s12part']), self.tpg, self._nv(), self.G, self.N, self.ldr, _lib.ptr(W['dy1']),
                      _lib.ptr(W['dy2']), gs, self.ldp, _lib.ptr(W['s12'][(k, 1)]), _lib.ptr(W['s12'][(k, 2)]),
                      _lib.ptr(W['coef'][0]), _lib.ptr(W['coef'][1]), st, tag='fgnn_chan_matmul_bwd16')
            hook('matmul_bwd', k)
            if self.PAIR_BWD and self.nvalid is None and not getattr(hook, 'per_mlp', False):
                self._mlp_bwd_pair(params, k, sin, din, emit=not first)
                hook('mlp2_bwd', k)
            else:
                self._mlp_bwd(params, k, 1, sin, None, W['dy1'], W['coef'][0], din, None, True, False)
                hook('mlp1_bwd', k)
                self._mlp_bwd(params, k, 2, sin, None, W['dy2'], W['coef'][1], din, None, True, False, emit=not first)
                hook('mlp2_bwd', k)
            if not first:
                _lib.call('fgnn_gn_bwd_coef_tiles_tpg', _lib.ptr(W['s12part']), _lib.ptr(self.nrm[(k - 1, 3)]), self._nv(),
                          self.G, 32, self.N, self.tpg, _lib.ptr(W['s12'][(k - 1, 3)]), _lib.ptr(W['coef'][2]), st)
                coef3 = W['coef'][2]
            dy = din
        keys = [(k, j) for k in range(1, K + 1) for j in (1, 2, 3)]
        if getattr(self, '_loss_pending', False):
            keys.append('loss')
            self._loss_pending = False
        for lo in range(0, len(keys), _lib.MAX_GRAD_JOBS):
            chunk = keys[lo:lo + _lib.MAX_GRAD_JOBS]
            jobs = (_lib.GradJob * len(chunk))()
            for i, kj in enumerate(chunk):
                if kj == 'loss':
                    jobs[i].wpart = self.pair_loss.data_ptr()
                    jobs[i].count = 1
                    jobs[i].out = self._loss_target.data_ptr()
                    jobs[i].rows = self.B * self.score_blocks
                    jobs[i].scale = 1.0 / self.total_nodes
                    if getattr(self, '_loss_scale_dev', None) is not None:       # 1 / sum(n) as a device scalar (forward(inv_nodes_dev=...))
                        jobs[i].scale_dev = self._loss_scale_dev.data_ptr()
                    continue
                rec = L.mlp[kj]
                jobs[i].wpart = W['wpart'][kj].data_ptr()
                jobs[i].count = rec['count']
                if kj in ((1, 1), (1, 2)) and W.get('struct_rows', 0):
                    jobs[i].rows = W['struct_rows']
                jobs[i].out = grads.data_ptr() + 4 * rec['off']
                jobs[i].s12 = W['s12'][kj].data_ptr()
                jobs[i].nrm = self.nrm[kj].data_ptr()
                jobs[i].dgn_w = grads.data_ptr() + 4 * rec['gn_w']
                jobs[i].dgn_b = grads.data_ptr() + 4 * rec['gn_b']
            _lib.call('fgnn_grad_finalize', jobs, len(chunk), W['nwg'], self.G, 32, st)
        return grads

    def export_decisions(self, on=True):
        """Test-only (tests/test_gpu_grad_pinned.py): from the next forward on, every fgnn_mlp_fwd16 launch is followed by its
        decision-exporting twin (fgnn_debug_mlp_fwd16_masks); relu_decisions() + self.idx are then all the discrete ReLU / arg-max
        decisions of a step (constant-size batches, generic block 1)."""
        self.decisions = {} if on else None

    def relu_decisions(self):
        """{(block, mlp, hidden layer): bool (G, 32, N, N)}: [pre-activation > 0] as the kernels' ReLU saw it."""
        L, G, N = self.layout, self.G, self.N
        out = {}
        bit = torch.arange(32, device=self.device, dtype=torch.int32)
        for (k, j), buf in self.decisions.items():
            w = buf.view(G, L.depth - 1, 32, self.tpg, 2)                                   # [..., parity], bit jp = element 64 t + 2 jp + parity
            m = ((w.unsqueeze(-1) >> bit) & 1)                                              # (G, d-1, 32, tpg, 2, 32)
            m = m.permute(0, 1, 2, 3, 5, 4).reshape(G, L.depth - 1, 32, self.tpg * 64)      # element order within the ldr-pitched plane
            m = m[..., :N * self.ldr].reshape(G, L.depth - 1, 32, N, self.ldr)[..., :N]
            for l in range(L.depth - 1):
                out[(k, j, l)] = m[:, l].bool()
        return out

    def step(self, params, grads, x, nvalid=None, total_nodes=None, loss_out=None, bits=None):
        scores, loss = self.forward(params, x, nvalid, total_nodes, defer_loss=True, loss_out=loss_out, bits=bits)
        self.backward(params, grads)
        return scores, loss

    # ------------------------------------------------------------------ inspection (tests)
    def load_dense(self, buf, t):
        """(G, C, N, N) fp32 tensor -> bf16 workspace slab `buf` (rounded to nearest even, padding zero-filled)."""
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        Cc = t.shape[1]
        _lib.call('fgnn_to_bf16', _lib.ptr(t), self._nv(), self.G, Cc, self.N, self.ldr, _lib.ptr(buf), Cc * self.ldp, self.ldp,
                  _lib.stream_ptr())

    def dense(self, buf, channels=32):
        """bf16 workspace slab -> (G, channels, N, N) fp32 tensor (copy)."""
        y = torch.empty(self.G, channels, self.N, self.N, dtype=torch.float32, device=self.device)
        _lib.call('fgnn_from_bf16', _lib.ptr(buf), channels * self.ldp, self.ldp, self.G, channels, self.N, self.ldr,
                  _lib.ptr(y), _lib.stream_ptr())
        return y
