"""Two half-batch engines on two HIP streams (opt-in: ``bench.py --chains 2``).

At the reference's batch size (32 pairs of N = 50, BASELINE.json configs[1]) every kernel of the fused step is a single
round of load -> compute -> store, and about a quarter of each launch is its prologue (one cold memory round trip), its
end-of-kernel imbalance and its reduction.  Graph pairs are independent (models/trainers.py:60-68: the two branches only
meet in the per-pair score matrix), so ``FgnnEngineDual`` splits the batch into two chains of pairs -- each a complete
``FgnnEngine`` (forward, loss, backward) on its own stream -- whose persistent MLP kernels take half of the CUs each
(``cu_share = 2``: at most 128 workgroups).  The two chains then run side by side on disjoint CUs and one chain's ramps
fall under the other chain's streaming.  Captured in a HIP graph the fork / join become graph edges.  (Ragged batches:
the work-balanced tile ranges are cut for the full grid, so the MLP kernels of a ragged chain ignore ``cu_share`` and run 256
workgroups; the two chains then overlap like any two kernels on two streams, not on disjoint CUs.)

Measured (tools/gpu_dual_probe.py, 1 x MI355X, B = 32): 0.894 -> 0.877 ms per step, i.e. 2 % -- every kernel still takes
what its per-CU work takes, only the HBM idle time of the ramps is shared -- which is why the single engine stays the
default.  A first variant with half-CU workgroups (both chains co-resident on every CU) was slower than the single engine
(1.02 ms): the fp32-MFMA kernels are issue-bound on their SIMD, a second kernel's wave on the same SIMD buys nothing.

The chains meet once per step: ONE ``fgnn_grad_finalize`` launch reduces the weight-gradient partials of both (their
partial buffers are the two halves of one allocation).  All reductions keep a fixed order, so results are bit-reproducible
run to run; scores are bit-identical to the single engine's, gradients differ by the association of the partial sums.
"""
import torch

from . import _lib
from .engine import FgnnEngine


class FgnnEngineDual:
    CU_SHARE = 2        # fgnn_mlp_fwd_args.cu_share of both chains: full-size workgroups on disjoint halves of the CUs

    def __init__(self, layout, G, N, device, ragged=False, mfma=None):
        if G % 2 or G < 4:
            raise RuntimeError('FgnnEngineDual: G = 2 * pairs with at least two pairs (got G = %d)' % G)
        self.layout, self.G, self.N, self.device = layout, G, N, device
        self.B = G // 2
        self.h = [(self.B + 1) // 2, self.B // 2]                   # pairs per chain
        self.lo = [0, self.h[0]]
        self.sub = [FgnnEngine(layout, 2 * h, N, device, ragged=ragged, cu_share=self.CU_SHARE, mfma=mfma) for h in self.h]
        self.x3 = self.sub[0].x3
        self.ragged = ragged
        self.side = torch.cuda.Stream(device=device)
        a, b = self.sub
        b._packs = a._packs                                         # one set of operand images, packed once per step
        f32 = dict(dtype=torch.float32, device=device)
        # outputs in pair order: chain 0 holds pairs [0, h0), chain 1 the rest
        self.scores = torch.empty(self.B, N, N, **f32)
        self.lse = torch.empty(self.B, N, **f32)
        rows = [e.B * e.score_blocks for e in self.sub]
        self.pair_loss = torch.empty(sum(rows), **f32)
        self.pair_rows = sum(rows)
        self.loss = torch.empty(1, **f32)
        a.scores, b.scores = self.scores[:self.h[0]], self.scores[self.h[0]:]
        a.lse, b.lse = self.lse[:self.h[0]], self.lse[self.h[0]:]
        a.pair_loss, b.pair_loss = self.pair_loss[:rows[0]], self.pair_loss[rows[0]:]
        # GraphNorm records of both chains in one buffer per MLP (the affine-gradient reduction runs over all graphs)
        for kj in list(a.nrm):
            joint = torch.empty((a.G + b.G) * 32 * 4, **f32)
            a.nrm[kj], b.nrm[kj] = joint[:a.G * 32 * 4], joint[a.G * 32 * 4:]
        self._bwd_joined = False
        self._x = None
        self._bits = None
        self._nv = None

    # ------------------------------------------------------------------ inputs in chain order
    def _split(self, t, out):
        """t = [side 1 (B) ; side 2 (B)] stacked on dim 0 -> out[c] = [side 1 of chain c ; side 2 of chain c]."""
        B = self.B
        for c in (0, 1):
            lo, h = self.lo[c], self.h[c]
            out[c][:h].copy_(t[lo:lo + h])
            out[c][h:].copy_(t[B + lo:B + lo + h])

    def stage_inputs(self, x=None, nvalid=None, bits=None):
        """Copy a stacked batch x = cat(x1, x2) (or its bit-packed adjacency) into the two chains' input buffers (loader
        work: four small copies).  step(params, grads, None) then runs on the staged inputs."""
        if (x is None) == (bits is None):
            raise RuntimeError('FgnnEngineDual.stage_inputs: exactly one of x / bits')
        src = x if x is not None else bits
        if src.shape[0] != self.G or not src.is_cuda:
            raise RuntimeError('FgnnEngineDual: expected %d stacked graphs on the GPU, got %s on %s' % (self.G, tuple(src.shape), src.device))
        cur = self._x if x is not None else self._bits
        if cur is None or cur[0].shape[1:] != src.shape[1:] or cur[0].dtype != src.dtype:
            cur = [torch.empty((2 * h,) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device) for h in self.h]
        self._split(src, cur)
        if x is not None:
            self._x, self._bits = cur, None
        else:
            self._x, self._bits = None, cur
        if (nvalid is None) != (not self.ragged):
            raise RuntimeError('FgnnEngineDual: ragged flag and nvalid argument disagree')
        if nvalid is not None:
            if self._nv is None:
                self._nv = [torch.empty(2 * h, dtype=torch.int32, device=self.device) for h in self.h]
            self._split(nvalid.to(torch.int32), self._nv)

    def _inputs(self, c):
        return (self._x[c] if self._x is not None else None, self._nv[c] if self._nv is not None else None,
                self._bits[c] if self._bits is not None else None)

    # ------------------------------------------------------------------ the two chains
    def _chains(self, fn):
        """fn(chain index, engine) on two streams: chain 0 on the current stream, chain 1 on the side stream."""
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        fn(0, self.sub[0])
        with torch.cuda.stream(self.side):
            fn(1, self.sub[1])
        cur.wait_stream(self.side)

    def _join_bwd(self):
        """Backward workspaces: the partial buffers that fgnn_grad_finalize reduces become halves of one allocation."""
        if self._bwd_joined:
            return
        a, b = self.sub
        Wa, Wb = a._alloc_bwd(), b._alloc_bwd()
        f32 = dict(dtype=torch.float32, device=self.device)
        for kj in list(Wa['s12']):
            na, nb = Wa['s12'][kj].numel(), Wb['s12'][kj].numel()
            joint = torch.empty(na + nb, **f32)
            Wa['s12'][kj], Wb['s12'][kj] = joint[:na], joint[na:]
            na = Wa['wpart'][kj].numel()
            joint = torch.empty(2 * na, **f32)
            Wa['wpart'][kj], Wb['wpart'][kj] = joint[:na], joint[na:]
        self._bwd_joined = True

    def _total_nodes(self, total_nodes):
        if total_nodes is not None:
            return float(total_nodes)
        if self._nv is None:
            return float(self.B * self.N)
        return float(sum(int(nv[:h].sum().item()) for nv, h in zip(self._nv, self.h)))

    def forward(self, params, x, nvalid=None, total_nodes=None, defer_loss=False, loss_out=None, bits=None):
        """As FgnnEngine.forward: (scores (B, N, N) in pair order, loss)."""
        if x is not None or bits is not None:
            self.stage_inputs(x, nvalid, bits)
        total = self._total_nodes(total_nodes)
        target = self.loss if loss_out is None else loss_out
        self.sub[0].pack_operands(params)
        self._chains(lambda c, e: e.forward(params, *self._inputs(c)[:2], total_nodes=total, defer_loss=True,
                                            loss_out=target, bits=self._inputs(c)[2], pack=False))
        self._after_forward(total, target, defer_loss)
        return self.scores, target

    def _after_forward(self, total, target, defer_loss):
        a, b = self.sub
        b._loss_pending = False                     # the joint loss sum is chain 0's finalize job
        a._loss_pending = bool(defer_loss)
        if not defer_loss:
            _lib.call('fgnn_sum_scale', _lib.ptr(self.pair_loss), self.pair_rows, 1, 1.0 / total, _lib.ptr(target), _lib.stream_ptr())

    def _finalize(self, grads):
        a, b = self.sub
        a.grad_finalize(grads, rows=a._bwd['nwg'] + self.sub[1]._bwd['nwg'], graphs=a.G + b.G, pair_rows=self.pair_rows)

    def backward(self, params, grads, grad_scale=1.0):
        self._join_bwd()
        self._chains(lambda c, e: e.backward(params, grads, grad_scale, finalize=False))
        self._finalize(grads)
        return grads

    def step(self, params, grads, x, nvalid=None, total_nodes=None, loss_out=None, bits=None):
        """One training step's model work (forward + loss + backward of both chains, one gradient reduction).
        x = None and bits = None: run on the inputs of the last stage_inputs()."""
        if x is not None or bits is not None:
            self.stage_inputs(x, nvalid, bits)
        self._join_bwd()
        total = self._total_nodes(total_nodes)
        target = self.loss if loss_out is None else loss_out
        self.sub[0].pack_operands(params)

        def chain(c, e):
            xc, nvc, bc = self._inputs(c)
            e.forward(params, xc, nvc, total_nodes=total, defer_loss=True, loss_out=target, bits=bc, pack=False)
            e.backward(params, grads, finalize=False)

        self._chains(chain)
        self._after_forward(total, target, True)
        self._finalize(grads)
        return self.scores, target

    # ------------------------------------------------------------------ inspection
    @property
    def E(self):
        """Node embeddings (G, 32, N) in the stacked order [side 1 ; side 2]."""
        a, b = self.sub
        return torch.cat([a.E[:a.B], b.E[:b.B], a.E[a.B:], b.E[b.B:]])
