"""Fused execution of the 2-FGNN hot path on one MI355X.

``FgnnEngine`` runs what ``Siamese_Node_Exp.forward`` + ``triplet_loss`` +
autograd do in the reference (models/trainers.py:60-68, models/blocks_emb.py:16-43,
toolbox/losses.py:20-34) as a fixed sequence of HIP kernels from
``libfgnn_hip.so``.  Both branches of the siamese pair share weights, so they are
stacked into one batch of G = 2B graphs.  Activations live in HBM as
(G, 32, ldp) slabs of *pre-norm* values plus a small GraphNorm record per (g, c);
every consumer normalises on load, ``cat`` is never materialised, and the hidden
MLP activations never leave the register file (they are recomputed in backward).

Parameters and their gradients are flat fp32 buffers laid out in the reference's
``state_dict`` order (models/utils.py:57-58), so data-parallel training needs a
single all-reduce of ``grads``.
"""
import ctypes as C
import os

import torch

from . import _lib

EPS = 1e-05


class ParamLayout:
    """Offsets of every tensor of the node embedder inside one flat buffer, in the
    reference's ``named_parameters`` order."""

    def __init__(self, original_features_num=2, num_blocks=4, in_features=32, out_features=32, depth_of_mlp=3):
        if in_features != _lib.FGNN_H or out_features != _lib.FGNN_H:
            raise RuntimeError('the fused engine is built for in_features = out_features = 32 (got %d, %d); other widths '
                               'run through the per-layer modules (layers.MlpBlock_Real)' % (in_features, out_features))
        if not 1 <= depth_of_mlp <= _lib.FGNN_MAX_DEPTH:
            raise RuntimeError('depth_of_mlp must be in 1..%d' % _lib.FGNN_MAX_DEPTH)
        if original_features_num not in (2, 32):
            raise RuntimeError('the fused engine is built for original_features_num = 2 or 32 (got %d); other widths run '
                               'through the per-layer modules (layers.MlpBlock_Real)' % original_features_num)
        self.c0 = original_features_num
        self.num_blocks = num_blocks
        self.depth = depth_of_mlp
        self.entries = []        # (name, offset, shape)
        self.mlp = {}            # (blk, j) -> dict(off=..., cin=..., conv_off=[...], gn_w=..., gn_b=...)
        off = 0
        last = original_features_num
        for blk in range(1, num_blocks + 1):
            for j, cin in ((1, last), (2, last), (3, last + 32)):
                pfx = 'ne_bm_block%d_mlp%d.' % (blk, j)
                rec = {'off': off, 'cin': cin, 'w': [], 'b': []}
                c = cin
                for i in range(depth_of_mlp):
                    self.entries.append((pfx + 'convs.%d.weight' % i, off, (32, c, 1, 1)))
                    rec['w'].append(off)
                    off += 32 * c
                    self.entries.append((pfx + 'convs.%d.bias' % i, off, (32,)))
                    rec['b'].append(off)
                    off += 32
                    c = 32
                self.entries.append((pfx + 'gn.weight', off, (1, 32, 1, 1)))
                rec['gn_w'] = off
                off += 32
                self.entries.append((pfx + 'gn.bias', off, (1, 32, 1, 1)))
                rec['gn_b'] = off
                off += 32
                rec['count'] = _lib.mlp_param_count(cin, depth_of_mlp)
                self.mlp[(blk, j)] = rec
            last = 32
        self.total = off

    def flatten(self, state_dict, device):
        flat = torch.empty(self.total, dtype=torch.float32, device=device)
        for name, off, shape in self.entries:
            key = name if name in state_dict else 'node_embedder.' + name
            t = state_dict[key]
            n = t.numel()
            flat[off:off + n].copy_(t.reshape(-1).to(torch.float32))
        return flat

    def init_flat(self, seed, device):
        """Random initial weights in the reference's scheme (models/layers.py:134-142, 63-66):
        xavier-uniform conv weights, zero conv biases, GraphNorm weight 1 / bias 0."""
        gen = torch.Generator().manual_seed(seed)
        flat = torch.zeros(self.total, dtype=torch.float32)
        for name, off, shape in self.entries:
            n = _numel(shape)
            if name.endswith('.weight') and '.convs.' in name:
                fan_out, fan_in = shape[0], shape[1]
                bound = (6.0 / (fan_in + fan_out)) ** 0.5
                flat[off:off + n] = (torch.rand(n, generator=gen) * 2 - 1) * bound
            elif name.endswith('gn.weight'):
                flat[off:off + n] = 1.0
        return flat.to(device)

    def unflatten(self, flat):
        return {name: flat[off:off + _numel(shape)].view(shape) for name, off, shape in self.entries}


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _round_up(x, m):
    return (x + m - 1) // m * m


class EngineCache:
    """Per-shape engines with a workspace budget: least recently used engines are dropped beyond `budget` bytes (at least
    one is always kept).  Shared by FgnnTrainer and by the module path (Network), so a stream of ragged shapes re-uses a
    bounded set of workspaces in both."""

    def __init__(self, budget):
        import collections
        self.budget = budget
        self._items = collections.OrderedDict()        # key -> (engine, bytes)

    @staticmethod
    def engine_bytes(G, N, num_blocks, elt=4):
        ldp = -(-N * N // 32) * 32
        return (4 * num_blocks + 5) * G * 32 * ldp * elt          # forward + backward activation slabs dominate

    def get(self, key, factory, nbytes):
        """-> (engine, evicted keys)."""
        hit = self._items.get(key)
        if hit is not None:
            self._items.move_to_end(key)
            return hit[0], []
        eng = factory()
        self._items[key] = (eng, nbytes)
        evicted = []
        while self.used() > self.budget and len(self._items) > 1:
            k, _ = self._items.popitem(last=False)
            evicted.append(k)
        return eng, evicted

    def used(self):
        return sum(b for _, b in self._items.values())

    def keys(self):
        return list(self._items)

    def clear(self):
        self._items.clear()

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(list(self._items))

    def values(self):
        return [e for e, _ in self._items.values()]


class FgnnEngine:
    """Workspace + launch sequence for a fixed (G, N) problem on the current device."""
    SKIP_PADDING_TILES = True     # ragged engines: fgnn_ragged_tile_ranges + tile skipping in fgnn_mlp_fwd / fgnn_mlp_bwd
    MM_ORDER = True               # ragged engines: longest-job-first order of the whole-matrix per-channel products
    PAIR_BWD = os.environ.get('FGNN_PAIR_BWD', '1') != '0'      # mlp1 + mlp2 backward of a block as one launch (fgnn_mlp_bwd_pair)
    # step(): scoring + loss + their backward as ONE launch (fgnn_score_ce_step, bit-identical to the two).  Off by default: measured in
    # the replayed cfg2 step at 13.7 us against 5.8 + 7.0 us for the two launches it replaces (profiles/archive/r05_c_graph_timeline.txt) --
    # every workgroup of a pair repeats the score matrix and the row log-sum-exps, and inside a graph a launch boundary costs nothing
    SCORE_STEP = os.environ.get('FGNN_SCORE_STEP', '0') != '0'
    # the step's operand packing as extra workgroups of the structured block 1's first launch (FGNN_PACK_IN_STRUCT=0: its own launch)
    PACK_IN_STRUCT = os.environ.get('FGNN_PACK_IN_STRUCT', '1') != '0'
    # round 6: the MLP kernels on 16-pixel tiles / v_mfma_f32_16x16x4_f32 (csrc/*_t16.hip) where they are built; FGNN_T16=0: the
    # 32-pixel kernels everywhere.  A comma list selects kernels: 'pair' (mlp1 + mlp2 backward), 'bwd' (mlp3 backward) -- the default --
    # and, opt-in, the forward: 'fwd3' (mlp3 forward: -1.2 us per launch; z bit-identical for the same inputs, but its statistics are
    # per 16-pixel half, so the GraphNorm records -- and with them a few ReLU decisions of the following blocks -- round differently
    # than in the 32-pixel forward: another, equally valid fp32 evaluation, which three batch-level golden gates tuned on the 32-pixel
    # evaluation do not absorb; off by default), 'fwd12' (mlp1 + mlp2 forward: measured 0.3 us SLOWER per launch than the 32-pixel
    # kernel, and its consumer -- the per-channel product that finalizes the statistics in its prologue -- 1.9 us slower)
    T16 = os.environ.get('FGNN_T16', 'pair,bwd')

    # default contraction of the MLP kernels (FGNN_MFMA=x3 selects the split-bf16 kernels where they are built)
    MFMA = os.environ.get('FGNN_MFMA', 'f32')

    # block 1 on bit-packed inputs: 'generic' = the MLP / per-channel-product kernels every block uses (bit-identical to the
    # dense-input step); 'structured' = csrc/block1_struct.hip (class tables, closed-form product, class sums in the backward:
    # same function, results equal to fp32 rounding).  FGNN_BLOCK1 overrides the default.
    BLOCK1 = os.environ.get('FGNN_BLOCK1', 'generic')

    def __init__(self, layout, G, N, device, ragged=False, cu_share=0, mfma=None, block1=None):
        """cu_share=2: the persistent MLP kernels take half of the CUs (fgnn_mlp_fwd_args.cu_share), for engines that run
        next to another one on a second stream (FgnnEngineDual).  Ragged engines with tile ranges ignore it (full grid).
        mfma: 'f32' = v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); 'x3' = the bf16 matrix cores through the exact
        three-way operand split of csrc/fgnn_x3.h (fp32 tensors, fp32 accumulation, error of an fp32 rounding per product;
        built for depth 3 and constant-size batches -- other cases use 'f32')."""
        _lib.load()
        self.T16 = type(self).T16           # (snapshot: the operand images below are packed for the kernel set chosen here)
        self.cu_share = int(cu_share)
        mfma = self.MFMA if mfma is None else mfma
        if mfma not in ('f32', 'x3'):
            raise ValueError("mfma must be 'f32' or 'x3' (got %r)" % (mfma,))
        self.x3 = (mfma == 'x3' and not ragged and layout.depth == 3 and layout.c0 in (2, 32))
        parts = os.environ.get('FGNN_X3_PARTS', 'fwd,pair').split(',')
        self.x3_fwd, self.x3_pair = self.x3 and 'fwd' in parts, self.x3 and 'pair' in parts
        block1 = self.BLOCK1 if block1 is None else block1
        if block1 not in ('generic', 'structured'):
            raise ValueError("block1 must be 'generic' or 'structured' (got %r)" % (block1,))
        # the structured block 1 applies to bit-packed inputs (embed(bits=...)), constant-size or ragged, N <= 256; anything else runs generic
        self.struct1 = (block1 == 'structured' and cu_share == 0
                        and bool(_lib.load().fgnn_block1_struct_supported(N, layout.depth, layout.c0)))
        self._struct = None
        self.decisions = None       # test-only, see export_decisions()
        self.layout = layout
        self.G, self.N = G, N
        self.P = N * N
        self.ldp = _round_up(self.P, 32)
        self.tpg = _lib.tiles_per_graph(N)
        self.device = device
        K = layout.num_blocks
        f32 = dict(dtype=torch.float32, device=device)
        act = lambda: torch.empty(G * 32 * self.ldp, **f32)
        self.z = {(k, j): act() for k in range(1, K + 1) for j in (1, 2, 3)}
        self.mult = {k: act() for k in range(1, K + 1)}
        self.nrm = {(k, j): torch.empty(G * 32 * 4, **f32) for k in range(1, K + 1) for j in (1, 2, 3)}
        # tile statistics of the forward MLP kernels: one record per 32-pixel tile, or per 16-pixel half with fgnn_mlp_fwd_t16
        self.part = [torch.empty(G * 2 * self.tpg * 32 * 2, **f32) for _ in range(2)]
        self.cnt = torch.empty(G * 2 * self.tpg, **f32)
        self.E = torch.empty(G, 32, N, **f32)
        self.idx = torch.empty(G, 32, N, dtype=torch.int32, device=device)
        self.B = G // 2
        self.scores = torch.empty(self.B, N, N, **f32)
        self.lse = torch.empty(self.B, N, **f32)
        self.score_blocks = _lib.load().fgnn_score_row_blocks(self.B, N)      # row blocks per pair of the scoring kernel
        self._score_step_ok = bool(_lib.load().fgnn_score_ce_step_supported(self.B, 32, N)) and self.score_blocks <= 256
        self.pair_loss = torch.empty(self.B * self.score_blocks, **f32)
        self.loss = torch.empty(1, **f32)
        self.nvalid = torch.empty(G, dtype=torch.int32, device=device) if ragged else None
        self._nvalid_own = self.nvalid      # the engine's own buffer; an int32 device tensor handed in is used in place (no copy launch)
        # ragged batches: work-balanced tile ranges of the MLP kernels (padding-only tiles are stepped over)
        self.ranges = (torch.empty(_lib.FGNN_RANGE_WG + 1, dtype=torch.int32, device=device)
                       if ragged and self.SKIP_PADDING_TILES else None)
        # ... and the largest-graph-first workgroup order of the whole-matrix per-channel products (64 < N <= 256), written
        # by the same launch
        self.mm_order = (torch.empty(G, dtype=torch.int32, device=device)
                         if self.ranges is not None and self.MM_ORDER and 64 < N <= 256 else None)
        # backward workspace (allocated lazily)
        self._bwd = None
        self.x = None
        self.xbits = None         # bit-packed adjacency input (embed(..., bits=...)) and its row sums
        self.xdeg = None
        # LDS operand images of every MLP launch, re-packed once per step (fgnn_pack_operands)
        self._packs = {}
        for k in range(1, K + 1):
            cin = layout.c0 if k == 1 else 32
            # image kinds: 0 / 1 = forward / backward image of the kernel set in use; with x3, kind + 2 = an fp32-MFMA image
            # packed by the same launch.  mlp3 stays on the fp32-MFMA kernels in BOTH directions: the two weight-gradient
            # slabs of its backward leave no registers for the split operands (measured 52-56 us against 48 us), and a
            # backward must recompute the hidden activations with the arithmetic of its forward (a ReLU mask that differs
            # from the forward's on a pre-activation within rounding distance of 0 is a gradient error of the flip class).
            fl = _lib.load().fgnn_pack_x3_floats if self.x3 else _lib.load().fgnn_pack_floats
            f3 = 2 if self.x3 else 0
            b3 = 3 if self.x3 else 1
            # (measurement switch FGNN_X3_PARTS = 'fwd' / 'pair': only that half of the x3 kernel pair, the other on fp32 MFMAs)
            f12 = 2 if (self.x3 and not self.x3_fwd) else 0
            b12 = 3 if (self.x3 and not self.x3_pair) else 1
            if self._t16_pair(cin):
                b12 = 5
            if self._t16_fwd(2):
                f12 = 4
            if self._t16_fwd(1):
                f3 = 4
            self._packs[('f', k, 12)] = (f12, cin, 0, 2, torch.empty(fl(f12, cin, 0, layout.depth, 2), **f32))
            self._packs[('f', k, 3)] = (f3, 32, cin, 1, torch.empty(fl(f3, 32, cin, layout.depth, 1), **f32))
            for j in (1, 2):
                self._packs[('b', k, j)] = (b12, cin, 0, 1, torch.empty(fl(b12, cin, 0, layout.depth, 1), **f32))
            if self._t16_bwd3(cin):
                b3 = 5
            self._packs[('b', k, 3)] = (b3, 32, cin, 1, torch.empty(fl(b3, 32, cin, layout.depth, 1), **f32))

    # ------------------------------------------------------------------ helpers
    def _t16(self, what):
        return self.T16 not in ('0', '') and what in self.T16.split(',')

    def _t16_pair(self, cin):
        """mlp1 + mlp2 backward of a block on the 16-pixel-tile kernel (fgnn_mlp_bwd_pair_t16): dense 32-channel input slab, depth 3"""
        return (self._t16('pair') and self.PAIR_BWD and not self.x3_pair and self.layout.depth == 3 and cin == 32 and self.N <= 256)

    def _t16_fwd(self, nmlp):
        """a forward MLP launch (nmlp = 2: mlp1 + mlp2, 1: mlp3) on the 16-pixel-tile kernel (fgnn_mlp_fwd_t16: statistics per 16-pixel half)"""
        return self._t16('fwd3' if nmlp == 1 else 'fwd12') and not self.x3 and self.layout.depth == 3 and self.N <= 256

    def _recs(self, nmlp):
        """statistics records per graph of the forward launch with nmlp MLPs"""
        return 2 * self.tpg if self._t16_fwd(nmlp) else self.tpg

    def _t16_bwd3(self, cin):
        """mlp3 backward of a block on the 16-pixel-tile kernel (fgnn_mlp_bwd_t16): depth 3, input [mult ; 32 or 2 channels]"""
        return self._t16('bwd') and not self.x3 and self.layout.depth == 3 and cin in (2, 32) and self.N <= 256

    def _nv(self):
        return _lib.ptr(self.nvalid) if self.nvalid is not None else None

    def _w(self, params, off):
        return params.data_ptr() + 4 * off

    def _slab_in(self, k, params):
        """Input slab of block k: raw x for k == 1, else block k-1's mlp3 output (normalise on load)."""
        if k == 1:
            if self.xbits is not None:      # never materialised: block 1's kernels expand the packed adjacency themselves
                s = _lib.Slab()
                s.gstride, s.ldp, s.C = self.layout.c0 * self.P, self.P, self.layout.c0
                return s
            return _lib.make_slab(self.x, self.layout.c0 * self.P, self.P, self.layout.c0)
        rec = self.layout.mlp[(k - 1, 3)]
        s = _lib.make_slab(self.z[(k - 1, 3)], 32 * self.ldp, self.ldp, 32, nrm=self.nrm[(k - 1, 3)])
        s.beta = self._w(params, rec['gn_b'])
        return s

    def _slab_z(self, k, j, params):
        rec = self.layout.mlp[(k, j)]
        s = _lib.make_slab(self.z[(k, j)], 32 * self.ldp, self.ldp, 32, nrm=self.nrm[(k, j)])
        s.beta = self._w(params, rec['gn_b'])
        return s

    def _slab_raw(self, t):
        return _lib.make_slab(t, 32 * self.ldp, self.ldp, 32)

    def pack_operands(self, params):
        """Pack the LDS operand images of all MLP launches of one step (one small launch)."""
        L = self.layout
        self._pack_launch(params, list(self._packs.items()), 'fgnn_pack_x3_operands' if self.x3 else 'fgnn_pack_operands')

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

    def _pack_launch(self, params, items, entry):
        for lo in range(0, len(items), _lib.MAX_PACK_JOBS):
            chunk = items[lo:lo + _lib.MAX_PACK_JOBS]
            _lib.call(entry, self._pack_jobs(params, chunk), len(chunk), _lib.stream_ptr())

    def _mlp_fwd(self, params, k, js, a, b, finalize=True):
        """finalize=False: leave the tile statistics of the two MLPs un-finalized (the matmul that consumes them
        finalizes them in its own prologue, fgnn_chan_matmul_fwd_fin)."""
        L = self.layout
        args = _lib.MlpFwdArgs()
        args.G, args.N, args.depth, args.nmlp = self.G, self.N, L.depth, len(js)
        args.nvalid = self.nvalid.data_ptr() if self.nvalid is not None else None
        args.a = a
        if b is not None:
            args.b = b
        for m, j in enumerate(js):
            rec = L.mlp[(k, j)]
            for l in range(L.depth):
                args.W[m][l] = self._w(params, rec['w'][l])
                args.bias[m][l] = self._w(params, rec['b'][l])
            args.z[m] = self.z[(k, j)].data_ptr()
            args.part[m] = self.part[m].data_ptr()
        args.ldz = self.ldp
        args.cnt = self.cnt.data_ptr()
        args.packed = self._packs[('f', k, 12 if len(js) == 2 else 3)][4].data_ptr()
        self._packed_input(args, k)
        if self.ranges is not None:
            args.ranges = self.ranges.data_ptr()
        args.cu_share = self.cu_share
        st = _lib.stream_ptr()
        entry = 'fgnn_mlp_fwd_x3' if (self.x3_fwd and len(js) == 2) else 'fgnn_mlp_fwd'
        recs = self._recs(len(js))
        if self._t16_fwd(len(js)):
            entry = 'fgnn_mlp_fwd_t16'
        _lib.call(entry, C.byref(args), st, tag='mlp_fwd[cin=%d,nmlp=%d]' % (a.C + (b.C if b is not None else 0), len(js)))
        if self.decisions is not None:
            # test-only (export_decisions): the decision-exporting twin of the launch above -- same tile code, same outputs written
            # once more -- leaves one bit per hidden pre-activation of these MLPs
            bufs = [torch.zeros(self.G * (L.depth - 1) * 32 * self.tpg, dtype=torch.int32, device=self.device) for _ in js]
            if entry == 'fgnn_mlp_fwd_t16':
                # the twin is the 32-pixel kernel: same inputs, same records, the same chain of fused multiply-adds -> the same decisions.
                # Its outputs go to scratch (its statistics records have another granularity), its image it builds itself (kind 0)
                scr = [torch.empty_like(self.z[(k, j)]) for j in js]
                spart = [torch.empty(self.G * self.tpg * 32 * 2, dtype=torch.float32, device=self.device) for _ in js]
                scnt = torch.empty(self.G * self.tpg, dtype=torch.float32, device=self.device)
                for m in range(len(js)):
                    args.z[m] = scr[m].data_ptr()
                    args.part[m] = spart[m].data_ptr()
                args.cnt = scnt.data_ptr()
                args.packed = None
                self._dbg_keep = (scr, spart, scnt)
            _lib.call('fgnn_debug_mlp_fwd_x3_masks' if (self.x3_fwd and len(js) == 2) else 'fgnn_debug_mlp_fwd_masks', C.byref(args),
                      _lib.ptr(bufs[0]), _lib.ptr(bufs[1]) if len(js) == 2 else None, st)
            for j, buf in zip(js, bufs):
                self.decisions[(k, j)] = buf
        if not finalize:
            return
        if len(js) == 2:
            r0, r1 = L.mlp[(k, js[0])], L.mlp[(k, js[1])]
            _lib.call('fgnn_gn_finalize2_r', _lib.ptr(self.part[0]), _lib.ptr(self.part[1]), _lib.ptr(self.cnt),
                      C.c_void_p(self._w(params, r0['gn_w'])), C.c_void_p(self._w(params, r1['gn_w'])), self._nv(),
                      self.G, 32, self.N, recs, EPS, _lib.ptr(self.nrm[(k, js[0])]), _lib.ptr(self.nrm[(k, js[1])]), st)
        else:
            rec = L.mlp[(k, js[0])]
            _lib.call('fgnn_gn_finalize_r', _lib.ptr(self.part[0]), _lib.ptr(self.cnt),
                      C.c_void_p(self._w(params, rec['gn_w'])), self._nv(), self.G, 32, self.N, recs, EPS,
                      _lib.ptr(self.nrm[(k, js[0])]), st)

    def _packed_input(self, args, k):
        """Block 1 with a bit-packed input: its 2-channel slab is expanded inside the kernel."""
        if k == 1 and self.xbits is not None:
            args.xbits = self.xbits.data_ptr()
            args.xdeg = self.xdeg.data_ptr()

    # ------------------------------------------------------------------ forward
    def embed(self, params, x, nvalid=None, bits=None, pack=True):
        """x: (G, c0, N, N) contiguous device tensor -- or bits: (G, N, ceil(N/32)) int32 words of the bit-packed
        adjacency (bit j of row i = W[i][j], the format of inputs.expand_adjacency / synthetic.pack_adjacency): the
        (2, N, N) representation of loaders/data_generator.py:118-125 is then built inside block 1's kernels and never
        exists in HBM.  Fills self.E / self.idx.
        LIFETIME: x / bits, and an nvalid handed over as a contiguous int32 device tensor of G entries, are read IN PLACE -- by this
        forward AND by the backward that follows it (block 1's backward re-reads the input; every backward kernel reads nvalid): the
        caller must leave them untouched until backward() has been issued (stream order is enough).  Any other nvalid (host list,
        int64, another device) is copied into the engine's own buffer.  A HIP graph captured around step() keeps those addresses."""
        L = self.layout
        if (nvalid is None) != (self.nvalid is None):
            raise RuntimeError('FgnnEngine: ragged flag and nvalid argument disagree')
        if nvalid is not None:
            if nvalid.dtype == torch.int32 and nvalid.is_cuda and nvalid.is_contiguous() and nvalid.numel() == self.G:
                self.nvalid = nvalid            # read in place by every kernel of the step (a copy node costs 4.6 + 8.6 us of gap in a replayed graph)
            else:
                self._nvalid_own.copy_(nvalid.to(torch.int32))
                self.nvalid = self._nvalid_own
        st = _lib.stream_ptr()
        if self.ranges is not None:
            _lib.call('fgnn_ragged_tile_ranges_order', _lib.ptr(self.nvalid), self.G, self.N, _lib.ptr(self.ranges),
                      _lib.ptr(self.mm_order) if self.mm_order is not None else None, st, tag='fgnn_ragged_tile_ranges')
        if bits is not None:
            words = (self.N + 31) // 32
            if x is not None or L.c0 != 2 or L.depth != 3:
                raise RuntimeError('FgnnEngine.embed: bits= replaces x and needs original_features_num = 2, depth_of_mlp = 3')
            if tuple(bits.shape) != (self.G, self.N, words) or bits.dtype not in (torch.int32, torch.uint32) \
                    or not bits.is_contiguous() or bits.device.type != 'cuda':
                raise RuntimeError('FgnnEngine.embed: expected contiguous 32-bit words %s on the GPU, got %s %s'
                                   % ((self.G, self.N, words), tuple(bits.shape), bits.dtype))
            if self.xdeg is None:
                self.xdeg = torch.empty(self.G * self.N, dtype=torch.float32, device=self.device)
            self.x, self.xbits = None, bits
            if not self.struct1:        # (the structured block 1 writes the row sums itself: one launch less)
                _lib.call('fgnn_adjacency_degree', _lib.ptr(bits), self._nv(), self.G, self.N, _lib.ptr(self.xdeg), st)
        else:
            if x.shape != (self.G, L.c0, self.N, self.N) or not x.is_contiguous() or x.dtype != torch.float32:
                raise RuntimeError('FgnnEngine.embed: expected contiguous fp32 %s, got %s %s'
                                   % ((self.G, L.c0, self.N, self.N), tuple(x.shape), x.dtype))
            self.x, self.xbits = x, None
        struct_now = self.struct1 and self.xbits is not None
        # the structured block 1's first launch carries the packing as extra workgroups (fgnn_block1_struct_fwd_pack: one launch less)
        pack_in_struct = pack and struct_now and self.PACK_IN_STRUCT and not self.x3 and len(self._packs) <= _lib.MAX_PACK_JOBS
        if pack and not pack_in_struct:     # (FgnnEngineDual packs once for both of its engines.  Running this launch on a second stream beside
            self.pack_operands(params)      # the structured block 1, which reads no operand image, was measured: +20 us per step)
        for k in range(1, L.num_blocks + 1):
            sin = self._slab_in(k, params)
            if k == 1 and struct_now:
                self._struct_fwd(params, with_pack=pack_in_struct)
                pool_fin = L.num_blocks == 1 and bool(_lib.load().fgnn_colmax_fwd_fin_supported(self.N))
                self._mlp_fwd(params, 1, (3,), self._slab_raw(self.mult[1]), sin, finalize=not pool_fin)
                continue
            # finalize-in-prologue lengthens every matmul workgroup by ~2 us: it beats the separate finalize launch
            # (~6 us) only while the matmul runs a few workgroup rounds (measured cross-over: B ~ 64-128 pairs at N = 50)
            fin = bool(_lib.load().fgnn_chan_matmul_fwd_fin_supported(self.N)) and self.G * 32 <= 4096
            self._mlp_fwd(params, k, (1, 2), sin, None, finalize=not fin)
            ya, yb = self._slab_z(k, 1, params), self._slab_z(k, 2, params)
            if fin:     # the matmul finalizes the GraphNorm records of mlp1 / mlp2 itself (one launch less)
                r1, r2 = L.mlp[(k, 1)], L.mlp[(k, 2)]
                _lib.call('fgnn_chan_matmul_fwd_fin_ord_r', C.byref(ya), C.byref(yb), _lib.ptr(self.part[0]), _lib.ptr(self.part[1]),
                          _lib.ptr(self.cnt), C.c_void_p(self._w(params, r1['gn_w'])), C.c_void_p(self._w(params, r2['gn_w'])),
                          EPS, self._nv(), self.G, self.N, self._recs(2), _lib.ptr(self.mult[k]), 32 * self.ldp, self.ldp,
                          _lib.ptr(self.mm_order) if self.mm_order is not None else None, self._fill(), st, tag='fgnn_chan_matmul_fwd')
            else:
                _lib.call('fgnn_chan_matmul_fwd_ord', C.byref(ya), C.byref(yb), self._nv(), self.G, self.N,
                          _lib.ptr(self.mult[k]), 32 * self.ldp, self.ldp,
                          _lib.ptr(self.mm_order) if self.mm_order is not None else None, self._fill(), st, tag='fgnn_chan_matmul_fwd')
            # the last block's statistics are finalized by the pooling kernel that consumes them
            pool_fin = k == L.num_blocks and bool(_lib.load().fgnn_colmax_fwd_fin_supported(self.N))
            self._mlp_fwd(params, k, (3,), self._slab_raw(self.mult[k]), sin, finalize=not pool_fin)
        out = self._slab_z(L.num_blocks, 3, params)
        if pool_fin:
            rec = L.mlp[(L.num_blocks, 3)]
            _lib.call('fgnn_colmax_fwd_fin_r', C.byref(out), _lib.ptr(self.part[0]), _lib.ptr(self.cnt),
                      C.c_void_p(self._w(params, rec['gn_w'])), EPS, self._nv(), self.G, self.N, self._recs(1), _lib.ptr(self.E),
                      _lib.ptr(self.idx), st, tag='fgnn_colmax_fwd')
        else:
            _lib.call('fgnn_colmax_fwd', C.byref(out), self._nv(), self.G, self.N, _lib.ptr(self.E), _lib.ptr(self.idx), st)
        return self.E

    def _fill(self):
        """Padding of the per-channel products' outputs: with tile ranges every consumer steps over padding-only tiles, so only what
        shares a tile with a valid pixel is zeroed (include/fgnn_hip.h, fgnn_chan_matmul_fwd_ord)."""
        return 1 if self.ranges is not None else 0

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
        """mlp1 + mlp2 + mult of block 1 from the class tables: two small launches instead of fgnn_mlp_fwd + fgnn_chan_matmul_fwd.
        with_pack: the first of them also packs the operand images of the step's MLP launches (pack_operands, without its launch)."""
        S = self._struct_ws()
        st = _lib.stream_ptr()
        (w1, b1), (w2, b2) = self._w3(params, 1), self._w3(params, 2)
        r1, r2 = self.layout.mlp[(1, 1)], self.layout.mlp[(1, 2)]
        args = [_lib.ptr(self.xbits), self._nv(), self.G, self.N, _lib.ptr(S['tab']),
                C.c_void_p(self._w(params, r1['gn_w'])), C.c_void_p(self._w(params, r1['gn_b'])),
                C.c_void_p(self._w(params, r2['gn_w'])), C.c_void_p(self._w(params, r2['gn_b'])), EPS,
                _lib.ptr(self.nrm[(1, 1)]), _lib.ptr(self.nrm[(1, 2)]), _lib.ptr(self.mult[1]), 32 * self.ldp, self.ldp,
                _lib.ptr(self.xdeg), _lib.ptr(S['ws']), w1, b1, w2, b2]      # (the class tables are built by the same launch)
        if with_pack:
            items = list(self._packs.items())
            _lib.call('fgnn_block1_struct_fwd_pack', *args, self._pack_jobs(params, items), len(items), st)
        else:
            _lib.call('fgnn_block1_struct_fwd', *args, st)

    def _struct_bwd(self, params):
        """... and in the backward direction: class sums of d(mult), then the per-class GraphNorm / conv backward, instead of
        fgnn_chan_matmul_bwd + fgnn_mlp_bwd_pair.  Leaves wpart / s12 of mlp1 and mlp2 for fgnn_grad_finalize."""
        S, W = self._struct_ws(), self._bwd
        # the structured backward writes the first fgnn_block1_struct_rows(G, N) partial rows of mlp1 / mlp2 of block 1; the reduction of
        # this step reads exactly those (grad_finalize below), whatever the other rows hold from a dense-input step
        W['struct_rows'] = int(_lib.load().fgnn_block1_struct_rows(self.G, self.N))
        (w1, _), (w2, _) = self._w3(params, 1), self._w3(params, 2)
        r1, r2 = self.layout.mlp[(1, 1)], self.layout.mlp[(1, 2)]
        _lib.call('fgnn_block1_struct_bwd', _lib.ptr(self.xbits), self._nv(), self.G, self.N, _lib.ptr(S['tab']), w1, w2,
                  _lib.ptr(self.nrm[(1, 1)]), _lib.ptr(self.nrm[(1, 2)]),
                  C.c_void_p(self._w(params, r1['gn_b'])), C.c_void_p(self._w(params, r2['gn_b'])),
                  _lib.ptr(W['dmult']), 32 * self.ldp, self.ldp, _lib.ptr(S['ws']),
                  _lib.ptr(W['wpart'][(1, 1)]), _lib.ptr(W['wpart'][(1, 2)]), _lib.ptr(W['s12'][(1, 1)]), _lib.ptr(W['s12'][(1, 2)]),
                  _lib.stream_ptr())

    def forward(self, params, x, nvalid=None, total_nodes=None, defer_loss=False, loss_out=None, bits=None, pack=True,
                with_score_bwd=False):
        """Siamese forward on the stacked batch x = cat(x1, x2) (or its bit-packed adjacency, see embed): returns
        (scores, loss).
        defer_loss: leave the final sum of the per-pair losses to the gradient-finalize launch of the
        following backward() (one launch less per training step); `loss` is valid after that.
        with_score_bwd (step()): the backward of loss * 1 follows at once -- scoring, the loss and their backward run as ONE launch
        where fgnn_score_ce_step is built for the shape (bit-identical to the two launches); backward() then starts at the pooling."""
        self.embed(params, x, nvalid, bits=bits, pack=pack)
        B, N = self.B, self.N
        st = _lib.stream_ptr()
        e1, e2 = self.E[:B], self.E[B:]
        if total_nodes is None:
            total_nodes = B * N if nvalid is None else int(nvalid[:B].sum().item())
        self.total_nodes = float(total_nodes)
        self._dE_done = False
        if with_score_bwd and self.SCORE_STEP and self._score_step_ok:
            W = self._alloc_bwd()
            self._set_gscale(1.0 / self.total_nodes)
            _lib.call('fgnn_score_ce_step', _lib.ptr(e1), _lib.ptr(e2), self._nv(), _lib.ptr(W['gscale']), B, 32, N, self.score_blocks,
                      _lib.ptr(self.scores), _lib.ptr(self.lse), _lib.ptr(self.pair_loss), _lib.ptr(W['dE'][:B]), _lib.ptr(W['dE'][B:]), st)
            self._dE_done = True
        else:
            _lib.call('fgnn_score_ce_fwd_blocks', _lib.ptr(e1), _lib.ptr(e2), self._nv(), B, 32, N, self.score_blocks,
                      _lib.ptr(self.scores), _lib.ptr(self.lse), _lib.ptr(self.pair_loss), st)
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
        act = lambda: torch.empty(self.G * 32 * self.ldp, **f32)
        nwg = _lib.load().fgnn_mlp_bwd_num_workgroups()
        # cu_share = 2: half of the CUs = half the workgroups = half the partial rows -- but only for constant-size batches:
        # with `ranges` (ragged engines) the kernels ignore cu_share and run the full grid, one row of wpart per workgroup
        if self.cu_share == 2 and self.ranges is None:
            nwg //= 2
        L = self.layout
        keys = [(k, j) for k in range(1, L.num_blocks + 1) for j in (1, 2, 3)]
        self._bwd = {
            'dE': torch.empty(self.G, 32, self.N, **f32),
            'dy': [act(), act()],
            'dmult': act(), 'dy1': act(), 'dy2': act(),
            # per-MLP GraphNorm-backward sums and workgroup partials live until the final
            # fgnn_grad_finalize launch
            's12': {kj: torch.empty(self.G * 32 * 2, **f32) for kj in keys},
            # (the structured block 1 writes one row per graph: the other rows of its two buffers stay zero)
            'wpart': {kj: torch.empty(nwg * L.mlp[kj]['count'], **f32)
                      for kj in keys},
            's12part': torch.empty(self.G * self.tpg * 32 * 2, **f32),
            'coef': [torch.empty(self.G * 32 * 4, **f32) for _ in range(3)],
            'nwg': nwg,
            'gscale': torch.empty(1, **f32),
        }
        return self._bwd

    def _coef(self, kj, slot):
        """coef[slot] <- dz coefficients of MLP kj from its summed s12."""
        W = self._bwd
        _lib.call('fgnn_gn_bwd_coef', _lib.ptr(W['s12'][kj]), _lib.ptr(self.nrm[kj]), self._nv(), self.G, 32, self.N,
                  _lib.ptr(W['coef'][slot]), None, None, _lib.stream_ptr())

    def _mlp_bwd(self, params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit=False, dx_strides=None):
        """coef: precomputed coefficient buffer; None -> derived in-kernel from s12[(k,j)] and nrm[(k,j)];
        'tiles' -> every workgroup sums the per-tile S1/S2 partials (s12part) of the graphs it touches itself.
        dx_strides: (graph stride, channel stride) of the dx tensors when they are not workspace slabs (the gradient with
        respect to the model input, a (G, c0, N, N) tensor)."""
        args = self._mlp_bwd_args(params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit, dx_strides)
        entry = 'fgnn_mlp_bwd_x3' if (self.x3 and j != 3) else 'fgnn_mlp_bwd'
        if j == 3 and b is not None and self._t16_bwd3(b.C) and _lib.load().fgnn_mlp_bwd_t16_supported(C.byref(args)):
            entry = 'fgnn_mlp_bwd_t16'
        elif self._packs[('b', k, j)][0] == 5:
            args.packed = None              # (the image of this MLP was packed for a 16-pixel kernel: the 32-pixel one builds its own)
        _lib.call(entry, C.byref(args), _lib.stream_ptr(),
                  tag='mlp_bwd[cin=%d,dx=%d]' % (a.C + (b.C if b is not None else 0),
                                                (a.C if dxa is not None else 0) + (b.C if (b is not None and dxb is not None) else 0)))

    def _mlp_bwd_pair(self, params, k, sin, din, emit):
        """mlp1 + mlp2 of block k in ONE launch (csrc/mlp_bwd_pair.hip; with mfma='x3' csrc/mlp_bwd_pair_x3.hip): the two waves of a
        SIMD take one MLP each and sum the gradient of the shared input in LDS; d_in is bit-identical to the two accumulating
        launches it replaces."""
        W = self._bwd
        a1 = self._mlp_bwd_args(params, k, 1, sin, None, W['dy1'], None, None, None, False, False, False, None)
        a2 = self._mlp_bwd_args(params, k, 2, sin, None, W['dy2'], None, din, None, True, False, emit, None)
        entry = 'fgnn_mlp_bwd_pair_x3' if self.x3_pair else ('fgnn_mlp_bwd_pair_t16' if self._t16_pair(sin.C) else 'fgnn_mlp_bwd_pair')
        _lib.call(entry, C.byref(a1), C.byref(a2), _lib.stream_ptr(),
                  tag='mlp_bwd_pair[cin=%d,dx=%d]' % (sin.C, sin.C if din is not None else 0))

    def _mlp_bwd_args(self, params, k, j, a, b, dy, coef, dxa, dxb, acc_a, acc_b, emit, dx_strides):
        L = self.layout
        W = self._bwd
        rec = L.mlp[(k, j)]
        gs = 32 * self.ldp
        args = _lib.MlpBwdArgs()
        args.G, args.N, args.depth = self.G, self.N, L.depth
        args.nvalid = self.nvalid.data_ptr() if self.nvalid is not None else None
        args.a = a
        if b is not None:
            args.b = b
        for l in range(L.depth):
            args.W[l] = self._w(params, rec['w'][l])
            args.bias[l] = self._w(params, rec['b'][l])
        args.dy, args.dgstride, args.ldd = dy.data_ptr(), gs, self.ldp
        args.z, args.zgstride, args.ldz = self.z[(k, j)].data_ptr(), gs, self.ldp
        if isinstance(coef, str):
            args.s12tiles = W['s12part'].data_ptr()
            args.s12_out = W['s12'][(k, j)].data_ptr()
            args.znrm = self.nrm[(k, j)].data_ptr()
        elif coef is not None:
            args.coef = coef.data_ptr()
        else:
            args.s12 = W['s12'][(k, j)].data_ptr()
            args.znrm = self.nrm[(k, j)].data_ptr()
        dgs, dld = (gs, self.ldp) if dx_strides is None else dx_strides
        if dxa is not None:
            args.dxa, args.dxa_gstride, args.dxa_ld = dxa.data_ptr(), (gs if (b is not None and dx_strides is not None) else dgs), \
                (self.ldp if (b is not None and dx_strides is not None) else dld)
        if dxb is not None:
            args.dxb, args.dxb_gstride, args.dxb_ld = dxb.data_ptr(), dgs, dld
        args.accumulate_a, args.accumulate_b = int(acc_a), int(acc_b)
        args.wpart = W['wpart'][(k, j)].data_ptr()
        args.packed = self._packs[('b', k, j)][4].data_ptr()
        self._packed_input(args, k)
        if self.ranges is not None:
            args.ranges = self.ranges.data_ptr()
        args.cu_share = self.cu_share
        if emit:
            args.s12part = W['s12part'].data_ptr()
        return args

    def backward(self, params, grads, grad_scale=1.0, finalize=True, gscale_dev=None):
        """Backward of loss*grad_scale after forward(); fills the flat `grads` buffer (finalize=False: everything but the
        last launch, see backward_from_dE).
        gscale_dev: a 1-element fp32 DEVICE tensor that holds grad_scale / total_nodes (replaces both): the normaliser of a ragged
        batch then never visits the host, and a captured step stays valid when the next batch has another node count."""
        W = self._alloc_bwd()
        B, N = self.B, self.N
        st = _lib.stream_ptr()
        if getattr(self, '_dE_done', False) and gscale_dev is None and grad_scale == 1.0:
            self._dE_done = False              # forward(with_score_bwd=True) already left d loss / d E in W['dE']
            return self.backward_from_dE(params, grads, W['dE'], finalize=finalize)
        self._dE_done = False
        gs_t = W['gscale']
        if gscale_dev is not None:
            gs_t = gscale_dev                  # read in place (a 1-element fp32 device tensor; no copy launch)
        else:
            self._set_gscale(grad_scale / self.total_nodes)
        e1, e2 = self.E[:B], self.E[B:]
        _lib.call('fgnn_score_ce_bwd', _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(self.scores), _lib.ptr(self.lse),
                  self._nv(), _lib.ptr(gs_t), B, 32, N, _lib.ptr(W['dE'][:B]), _lib.ptr(W['dE'][B:]), st)
        return self.backward_from_dE(params, grads, W['dE'], finalize=finalize)

    def _set_gscale(self, gs):
        W = self._bwd
        if W.get('gscale_value') != gs:        # a 1-element fill kernel per step otherwise
            W['gscale'].fill_(gs)
            W['gscale_value'] = gs

    def backward_from_dE(self, params, grads, dE, finalize=True, dx=None):
        """Backward of the node embedder given d loss / d E  (G, 32, N).
        dx: optional zero-initialised (G, c0, N, N) fp32 tensor that receives the gradient with respect to the input x (the
        reference's autograd gives it whenever the input requires grad; dense inputs and the fp32-MFMA kernels only).

        Per block (last to first):  mlp3 bwd -> matmul bwd (+ S1/S2 of mlp1, mlp2) -> mlp1 bwd ->
        mlp2 bwd (accumulates d_in and emits the S1/S2 tile partials of the previous block's
        mlp3).  No separate reduction pass over the activations is needed; all parameter
        gradients are finished by ONE fgnn_grad_finalize launch at the end (finalize=False leaves that launch to the
        caller: grad_finalize(grads, [engines]) reduces the partials of several engines at once)."""
        L = self.layout
        W = self._alloc_bwd()
        st = _lib.stream_ptr()
        gs = 32 * self.ldp
        K = L.num_blocks
        # dz coefficients of mlp3 (blocks < K): summed from the tile partials inside the consumer's prologue when a
        # workgroup's tile range spans few graphs, else by a separate fgnn_gn_bwd_coef_tiles launch
        in_prologue = bool(_lib.load().fgnn_mlp_bwd_coef_tiles_supported(self.G, self.N)) and self.ranges is None
        dy = W['dy'][0]
        out = self._slab_z(K, 3, params)
        _lib.call('fgnn_colmax_bwd', _lib.ptr(dE), _lib.ptr(self.idx), self._nv(), self.G, 32, self.N,
                  _lib.ptr(dy), gs, self.ldp, C.byref(out), _lib.ptr(W['s12'][(K, 3)]), st)
        for k in range(K, 0, -1):
            sin = self._slab_in(k, params)
            first = (k == 1)
            din = None if first else W['dy'][(K - k + 1) % 2]
            dxs = None
            if first and dx is not None:
                if self.xbits is not None or self.x3 or tuple(dx.shape) != (self.G, L.c0, self.N, self.N) or dx.dtype != torch.float32 \
                        or not dx.is_contiguous():
                    raise RuntimeError('FgnnEngine: the input gradient needs a dense fp32 input, the fp32-MFMA kernels (mfma="f32") '
                                       'and a contiguous (G, c0, N, N) fp32 dx')
                din, dxs = dx, (L.c0 * self.P, self.P)
            # mlp3: inputs [mult ; in].  Last block: dz coefficients derived in-kernel from the pooling's
            # S1/S2; other blocks: from the tile partials summed by fgnn_gn_bwd_coef_tiles below.
            self._mlp_bwd(params, k, 3, self._slab_raw(self.mult[k]), sin, dy,
                          None if k == K else ('tiles' if in_prologue else W['coef'][2]), W['dmult'], din, False, False, dx_strides=dxs)
            if first and self.struct1 and self.xbits is not None and dxs is None:
                self._struct_bwd(params)
                dy = din
                continue
            ya, yb = self._slab_z(k, 1, params), self._slab_z(k, 2, params)
            _lib.call('fgnn_chan_matmul_bwd_ord', C.byref(ya), C.byref(yb), _lib.ptr(W['dmult']), gs, self.ldp,
                      self._nv(), self.G, self.N, _lib.ptr(W['dy1']), _lib.ptr(W['dy2']), gs, self.ldp,
                      _lib.ptr(W['s12'][(k, 1)]), _lib.ptr(W['s12'][(k, 2)]),
                      _lib.ptr(self.mm_order) if self.mm_order is not None else None, self._fill(), st, tag='fgnn_chan_matmul_bwd')
            if first:
                W['struct_rows'] = 0                  # block 1 ran the generic kernels: all partial rows are live
            if self.PAIR_BWD and dxs is None and L.depth == 3 and sin.C in (2, 32):
                self._mlp_bwd_pair(params, k, sin, din, emit=not first)
            else:
                self._mlp_bwd(params, k, 1, sin, None, W['dy1'], None, din, None, True, False, dx_strides=dxs)
                self._mlp_bwd(params, k, 2, sin, None, W['dy2'], None, din, None, True, False, emit=not first, dx_strides=dxs)
            if not first and not in_prologue:
                _lib.call('fgnn_gn_bwd_coef_tiles', _lib.ptr(W['s12part']), _lib.ptr(self.nrm[(k - 1, 3)]), self._nv(),
                          self.G, 32, self.N, _lib.ptr(W['s12'][(k - 1, 3)]), _lib.ptr(W['coef'][2]), st)
            dy = din
        if finalize:
            self.grad_finalize(grads)
        return grads

    def grad_finalize(self, grads, rows=None, graphs=None, pair_rows=None):
        """ONE launch: reduce the workgroup partials + GraphNorm affine gradients of all MLPs (and the deferred loss sum).
        rows / graphs / pair_rows: FgnnEngineDual reduces the partials of both of its engines at once -- their wpart, s12,
        nrm and pair_loss buffers are consecutive halves of one allocation, this engine holding the first."""
        L = self.layout
        W = self._bwd
        K = L.num_blocks
        st = _lib.stream_ptr()
        keys = [(k, j) for k in range(1, K + 1) for j in (1, 2, 3)]
        if getattr(self, '_loss_pending', False):
            keys.append('loss')
            self._loss_pending = False
        for lo in range(0, len(keys), _lib.MAX_GRAD_JOBS):
            chunk = keys[lo:lo + _lib.MAX_GRAD_JOBS]
            jobs = (_lib.GradJob * len(chunk))()
            for i, kj in enumerate(chunk):
                if kj == 'loss':        # loss = sum(pair_loss) / nodes rides along as one more reduction job
                    jobs[i].wpart = self.pair_loss.data_ptr()
                    jobs[i].count = 1
                    jobs[i].out = self._loss_target.data_ptr()
                    jobs[i].rows = self.B * self.score_blocks if pair_rows is None else pair_rows
                    jobs[i].scale = 1.0 / self.total_nodes
                    if getattr(self, '_loss_scale_dev', None) is not None:       # 1 / sum(n) as a device scalar (forward(inv_nodes_dev=...))
                        jobs[i].scale_dev = self._loss_scale_dev.data_ptr()
                    continue
                rec = L.mlp[kj]
                jobs[i].wpart = W['wpart'][kj].data_ptr()
                jobs[i].count = rec['count']
                if kj in ((1, 1), (1, 2)) and W.get('struct_rows', 0):
                    jobs[i].rows = W['struct_rows']         # block 1 on its structured input: the rows its backward wrote
                jobs[i].out = grads.data_ptr() + 4 * rec['off']
                jobs[i].s12 = W['s12'][kj].data_ptr()
                jobs[i].nrm = self.nrm[kj].data_ptr()
                jobs[i].dgn_w = grads.data_ptr() + 4 * rec['gn_w']
                jobs[i].dgn_b = grads.data_ptr() + 4 * rec['gn_b']
            _lib.call('fgnn_grad_finalize', jobs, len(chunk), W['nwg'] if rows is None else rows,
                      self.G if graphs is None else graphs, 32, st)
        return grads

    def step(self, params, grads, x, nvalid=None, total_nodes=None, loss_out=None, bits=None):
        """One training step's model work: forward + loss + backward.  (x / bits / an int32 device nvalid are read in place by both
        passes: see embed().)"""
        scores, loss = self.forward(params, x, nvalid, total_nodes, defer_loss=True, loss_out=loss_out, bits=bits, with_score_bwd=True)
        self.backward(params, grads)
        return scores, loss

    # ------------------------------------------------------------------ inspection (tests / module API)
    def export_decisions(self, on=True):
        """Test-only (tests/test_gpu_grad_pinned.py): from the next forward on, every fgnn_mlp_fwd launch is followed by its
        decision-exporting twin (fgnn_debug_mlp_fwd_masks) and `relu_decisions()` returns the ReLU masks the step took; together with
        self.idx (the arg-max of the pooling) these are ALL the discrete decisions of a step."""
        self.decisions = {} if on else None

    def relu_decisions(self, params=None):
        """{(block, mlp, hidden layer): bool (G, 32, N, N)}: [pre-activation > 0] as the kernels' ReLU saw it.  MLPs that ran on the
        class tables of the structured block 1 take theirs from the table (a class's hidden value > 0), expanded over the pixels."""
        L, G, N, P = self.layout, self.G, self.N, self.P
        out = {}
        bit = torch.arange(32, device=self.device, dtype=torch.int32)
        for (k, j), buf in self.decisions.items():
            w = buf.view(G, L.depth - 1, 32, self.tpg)
            m = ((w.unsqueeze(-1) >> bit) & 1).reshape(G, L.depth - 1, 32, self.tpg * 32)[..., :P].reshape(G, L.depth - 1, 32, N, N)
            for l in range(L.depth - 1):
                out[(k, j, l)] = m[:, l].bool()
        if self.struct1 and self.xbits is not None and (1, 1) not in self.decisions:
            nc = 2 + 2 * (N + 1)
            tab = self._struct['tab'].view(2, nc, 4, 32)
            words = (N + 31) // 32
            b = self.xbits.view(G, N, words).to(torch.int64) & 0xffffffff
            col = torch.arange(N, device=self.device)
            wmat = ((b[:, :, col // 32] >> (col % 32)) & 1)                       # (G, N, N) 0/1
            if self.nvalid is not None:
                nv = self.nvalid.to(torch.int64)
                inside = (col[None, :] < nv[:, None])
                wmat = wmat * (inside[:, :, None] & inside[:, None, :])
            deg = wmat.sum(-1)
            eye = torch.eye(N, device=self.device, dtype=torch.bool)
            diag_cls = 2 + 2 * deg + torch.diagonal(wmat, dim1=1, dim2=2)
            cls = torch.where(eye[None], diag_cls[:, :, None].expand(G, N, N), wmat)      # (G, N, N) class of every pixel
            for j in (1, 2):
                for l in range(2):
                    hv = tab[j - 1, :, l, :] > 0                                   # (classes, 32)
                    out[(1, j, l)] = hv[cls].permute(0, 3, 1, 2).contiguous()      # (G, 32, N, N)
        return out


    def normalized(self, k, j, params):
        """Materialise the normalised output of MLP (k, j) as a (G, 32, N, N) tensor."""
        rec = self.layout.mlp[(k, j)]
        y = torch.empty(self.G, 32, self.N, self.N, dtype=torch.float32, device=self.device)
        _lib.call('fgnn_gn_apply', _lib.ptr(self.z[(k, j)]), 32 * self.ldp, self.ldp, _lib.ptr(self.nrm[(k, j)]),
                  C.c_void_p(self._w(params, rec['gn_b'])), self._nv(), self.G, 32, self.N,
                  _lib.ptr(y), 32 * self.P, self.P, _lib.stream_ptr())
        return y

    def unpadded(self, buf):
        """View a (G*32*ldp) workspace slab as (G, 32, N, N) (copy)."""
        return buf.view(self.G, 32, self.ldp)[:, :, :self.P].reshape(self.G, 32, self.N, self.N).clone()
