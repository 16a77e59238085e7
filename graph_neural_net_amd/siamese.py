"""``Siamese_Node_Exp`` (models/trainers.py:20-68) without the Lightning shell: shared node
embedder on both graphs of a pair, scores[b,i,j] = <E1[b,:,i], E2[b,:,j]>.

The two branches share weights, so they are embedded as ONE stacked batch of 2B graphs
through the fused engine; the outer-product scoring and its backward are HIP kernels too.
Inputs arrive as the reference's loaders produce them: ``{'input': T}`` dicts
(loaders/loaders.py:12-15) or bare tensors / MaskedTensors.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .blocks import block, block_emb, node_embedding
from .losses import triplet_loss
from .metrics import accuracy_linear_assignment, accuracy_max
from .masked import MaskedTensor
from .network import Network

get_node_emb = {'node_embedding': node_embedding}
get_block_init = {'block_emb': block_emb}
get_block_inside = {'block': block}


class _ScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e1, e2, nvalid):
        e1, e2 = e1.contiguous(), e2.contiguous()
        B, Cc, N = e1.shape
        scores = torch.empty(B, N, N, dtype=torch.float32, device=e1.device)
        _lib.call('fgnn_score_ce_fwd', _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(nvalid) if nvalid is not None else None,
                  B, Cc, N, _lib.ptr(scores), None, None, _lib.stream_ptr())
        ctx.save_for_backward(e1, e2, nvalid)
        return scores

    @staticmethod
    def backward(ctx, ds):
        e1, e2, nvalid = ctx.saved_tensors
        B, Cc, N = e1.shape
        d1, d2 = torch.empty_like(e1), torch.empty_like(e2)
        _lib.call('fgnn_score_bwd', _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(ds.contiguous()),
                  _lib.ptr(nvalid) if nvalid is not None else None, B, Cc, N, _lib.ptr(d1), _lib.ptr(d2),
                  _lib.stream_ptr())
        return d1, d2, None


def _unwrap_input(x):
    return x['input'] if isinstance(x, dict) else x


class Siamese_Node_Exp(nn.Module):
    MODULE_STEPS_MAX = 8        # captured module steps kept per model (fused_step through the module surface), LRU

    # MaskedTensor batches run padded to a multiple of this (batches whose largest graphs fall in the same granule replay ONE graph; cfg5
    # through fused_step: 0.861 ms at 8, 0.876 at 16, engine on the exact size 0.846 -- profiles/archive/r05_cfg5_surface.txt)
    RAGGED_GRANULE = int(__import__('os').environ.get('FGNN_RAGGED_GRANULE', '8'))
    INPUT_CHECK_EVERY = 128     # input_form='tensor_representation': the device verdict is read back on every k-th fused_step (0: never)

    def __init__(self, original_features_num, node_emb, lr=1e-3, scheduler_decay=0.5, scheduler_step=3, lr_stop=1e-5,
                 metric=None, precision='fp32', input_form='dense'):
        """Same positional signature as the reference (models/trainers.py:21).  Three keyword-only extras:
        input_form: 'dense' (default) -- block 1 makes no assumption about the input tensor; 'tensor_representation' -- the caller
                states that every batch is what the reference's loaders yield (loaders/data_generator.py:118-125: channel 0 a 0/1
                adjacency, channel 1 = diag(row sums)), and `fused_step` / `FgnnTrainer.from_module` then run block 1 on that structure
                (csrc/block1_struct.hip: the batch is bit-packed on the device, mlp1 / mlp2 become class tables, the per-channel
                product a closed form; same function, results equal to fp32 rounding).  The statement is VERIFIED on the device by
                the packing kernel of every step; the verdict is read on the first step of each batch shape, on every
                INPUT_CHECK_EVERY-th step and by `check_input_form()`, and a batch that is not a tensor representation raises.
                Also settable later: `model.node_embedder.input_form = 'tensor_representation'`.
        metric: None -> the reference's default, the Hungarian matching accuracy (`accuracy_linear_assignment`: trainers.py:52,
                metrics.py:92-116; SciPy's assignment computed on the device, csrc/lsap.hip); 'max' -> the arg-max accuracy
                (`accuracy_max`, metrics.py:118-141); or any callable.  Neither copies the scores to the host.
        precision: 'fp32' (default) or 'bf16' (the bf16 engine; what `node_embedder.half()` also selects)."""
        super().__init__()
        node_emb = dict(node_emb)
        try:
            node_emb_type = get_node_emb[node_emb['type']]
        except KeyError:
            raise NotImplementedError(f"node embedding {node_emb['type']} is not implemented")
        try:
            node_emb['block_inside'] = get_block_inside[node_emb['block_inside']]
        except KeyError:
            raise NotImplementedError(f"block inside {node_emb['block_inside']} is not implemented")
        try:
            node_emb['block_init'] = get_block_init[node_emb['block_init']]
        except KeyError:
            raise NotImplementedError(f"block init {node_emb['block_init']} is not implemented")
        self.out_features = node_emb['out_features']
        self.node_embedder_dic = {'input': (None, []), 'ne': node_emb_type(original_features_num, **node_emb)}
        self.node_embedder = Network(self.node_embedder_dic)
        self.loss = triplet_loss()
        if metric is None or metric == 'linear_assignment':
            self.metric = accuracy_linear_assignment
        elif metric == 'max':
            self.metric = accuracy_max
        elif callable(metric):
            self.metric = metric
        else:
            raise ValueError('metric must be None, "linear_assignment", "max" or a callable (got %r)' % (metric,))
        if precision not in ('fp32', 'bf16'):
            raise ValueError('precision must be "fp32" or "bf16" (got %r)' % (precision,))
        if precision == 'bf16':
            self.node_embedder.half()
        if input_form not in ('dense', 'tensor_representation'):
            raise ValueError('input_form must be "dense" or "tensor_representation" (got %r)' % (input_form,))
        self.node_embedder.input_form = input_form
        self.lr, self.scheduler_decay, self.scheduler_step, self.lr_stop = lr, scheduler_decay, scheduler_step, lr_stop

    def forward(self, x1, x2):
        """x1, x2: (bs, features, n, n) (or MaskedTensors of ragged graphs, both sides sharing the
        per-pair vertex counts) -> raw scores (bs, n, n)."""
        x1, x2 = _unwrap_input(x1), _unwrap_input(x2)
        ragged = isinstance(x1, MaskedTensor)
        if ragged:
            t = torch.cat([x1.tensor, x2.tensor])
            stacked = MaskedTensor(t, torch.cat([x1.nvalid, x2.nvalid]), x1.masked_dims, x1.base_name)
            nvalid = x1.nvalid
        else:
            stacked = torch.cat([x1, x2])
            nvalid = None
        e = self.node_embedder.fused_embedding(stacked)
        et = e.tensor if ragged else e
        B = et.shape[0] // 2
        scores = _ScoreFn.apply(et[:B], et[B:], nvalid)
        if ragged:
            return MaskedTensor(scores, nvalid, (1, 2), x1.base_name)
        return scores

    def fused_step(self, x1, x2, capture=True, metric=False):
        """Forward of both branches + scoring + `triplet_loss` + the full backward (+ the step's metric, models/trainers.py:70-76)
        as the fused launch sequence of engine.FgnnEngine -- with capture=True ONE replayed HIP graph per (padded) batch shape,
        the step `bench.py` measures (the eager module path `loss = model.loss(model(x1, x2)); loss.backward()` of
        models/trainers.py:60-76 issues the same kernels but pays ~40 host launches and the autograd bookkeeping per step).
        Afterwards every trainable parameter's `.grad` is the gradient of this batch (views of one flat buffer, OVERWRITTEN
        per call like after `zero_grad()`), so any torch optimizer / scheduler steps as usual.

        Takes whatever the reference's loaders yield (loaders/loaders.py:5-15): tensors or {'input': T} dicts of constant-size
        batches, and MaskedTensor batches (both sides sharing the per-pair vertex counts).  A ragged batch runs zero-padded to its
        largest graph rounded up to a multiple of RAGGED_GRANULE = 8 (so batches of nearby sizes replay the same graph); the vertex counts and
        the loss normaliser 1 / sum(n) (toolbox/losses.py:27-34) live in device buffers that each call overwrites -- nothing
        about a batch is baked into the captured graph but its padded shape.  Models narrower than the engine (widths below 32,
        1 or 3..31 input channels) run on zero-padded parameters (Network._padded_layout): the scatter of the parameters and the
        gather of their gradients are part of the graph.  Widths / depths the engine is not built for (in_features > 32,
        depth_of_mlp > 3 ...) run the module path's own per-layer launch sequence, captured the same way.

        metric=True appends the model's matching accuracy (the default Hungarian accuracy or 'max', both device kernels) to the
        step: returns (loss, scores, (n_correct, n_vertices)) with the counts as DEVICE tensors -- no host synchronisation
        unless the caller reads them.  Returns (loss, raw scores) otherwise; scores are (bs, n, n) or a MaskedTensor; all outputs
        are device tensors that the NEXT call of the same shape overwrites."""
        x1, x2 = _unwrap_input(x1), _unwrap_input(x2)
        net = self.node_embedder
        lay = net._standard_layout()
        ragged = isinstance(x1, MaskedTensor)
        t1, t2 = (x1.tensor.rename(None), x2.tensor.rename(None)) if ragged else (x1, x2)
        if not t1.is_cuda:
            raise RuntimeError('graph_neural_net_amd only runs on the GPU (input is on %s)' % (t1.device,))
        builtin_metric = self.metric is accuracy_linear_assignment or self.metric is accuracy_max
        if lay is None or (ragged and self.loss.loss_reduction != 'mean'):
            # widths / depths the fused engine is not built for (or a per-graph loss weighting the engine's scoring kernel does not
            # have): the module path's own launch sequence, captured once per batch shape and replayed (constant-size batches)
            if ragged:
                raise RuntimeError('fused_step: MaskedTensor batches need the standard node_embedding graph and loss_reduction="mean"; '
                                   'use the module path `loss = model.loss(model(x1, x2)); loss.backward()`')
            out = self._captured_module_step(t1, t2, capture)
            if metric:
                acc, n = self.metric(out[1])
                return out + ((acc, n),)
            return out
        net._bind_flat()
        dev = t1.device
        B, nmax = t1.shape[0], t1.shape[-1]
        gran = max(1, int(self.RAGGED_GRANULE))
        N = -(-nmax // gran) * gran if ragged else nmax
        pad = net._pad
        c0e = pad['c0p'] if pad is not None else t1.shape[1]
        if tuple(t1.shape) != tuple(t2.shape):
            raise RuntimeError('fused_step: the two sides of the batch have different (padded) shapes %s / %s; use the module path '
                               '`loss = model.loss(model(x1, x2)); loss.backward()`' % (tuple(t1.shape), tuple(t2.shape)))
        # input_form='tensor_representation': the loader's dense batch is bit-packed (and verified) on the device and block 1 runs on
        # its structured form -- where the kernels are built for the shape (2 input channels, depth 3, N <= 256); else the dense path
        want_tr = getattr(net, 'input_form', 'dense') == 'tensor_representation' and t1.shape[1] == 2 and lay.c0 == 2 \
            and t1.dtype == torch.float32
        eng = net._engine_for_shape(2 * B, N, ragged, dev, 'step', block1='structured' if want_tr else None)
        tr = want_tr and bool(getattr(eng, 'struct1', False))
        st = getattr(eng, '_step_state', None)
        if st is None:
            st = eng._step_state = {'x': None if tr else torch.zeros(2 * B, c0e, N, N, dtype=torch.float32, device=dev), 'graph': {},
                                    'nv': torch.zeros(2 * B, dtype=torch.int32, device=dev) if ragged else None,
                                    'inv': torch.ones(1, dtype=torch.float32, device=dev),
                                    'correct': torch.zeros(B, dtype=torch.int32, device=dev), 'nmax': N, 'calls': 0}
            if tr:      # (the verdict flag is the model's, shared by all batch shapes: check_input_form() reads one word)
                st['bits'] = torch.zeros(2 * B, N, (N + 31) // 32, dtype=torch.int32, device=dev)
                st['flag'] = self.__dict__.setdefault('_tr_flags', {}).setdefault(dev, torch.zeros(1, dtype=torch.int32, device=dev))
        c0 = t1.shape[1]
        first_of_shape = st['calls'] == 0
        st['calls'] += 1
        if ragged:
            if first_of_shape or (self.INPUT_CHECK_EVERY and st['calls'] % self.INPUT_CHECK_EVERY == 0):
                # both sides must share the per-pair vertex counts (the loss has target arange(n): toolbox/losses.py:27-34)
                if not torch.equal(x1.nvalid, x2.nvalid):
                    raise RuntimeError('fused_step: the two sides of a MaskedTensor batch must share the per-pair vertex counts; use '
                                       'the module path `loss = model.loss(model(x1, x2)); loss.backward()`')
            # Between those checks side 2 runs with side 1's counts (as FgnnTrainer does): a batch whose sides differ then still has the
            # reference's target / normaliser semantics for side 1 instead of silently mixing two sets of counts
            if not tr:
                st['nv'][:B].copy_(x1.nvalid)
                st['nv'][B:].copy_(x1.nvalid)
        if tr:
            # ONE launch straight from the loader's tensors, both sides: rows of ballots -> (2 B, N, ceil(N/32)) words of the engine's padded
            # size, with the verdict (channel 0 in {0, 1}, channel 1 = diag(row sums), counts within the padded size) OR-ed into a flag;
            # ragged batches: the same launch copies the vertex counts into the engine's buffer and leaves 1 / sum(n) in st['inv']
            nv1 = nv2 = None
            if ragged:
                nv1, nv2 = x1.nvalid, x1.nvalid            # (side 2 with side 1's counts, see above)
                if nv1.dtype != torch.int32 or not nv1.is_cuda or not nv1.is_contiguous() or nv2.dtype != torch.int32 \
                        or not nv2.is_cuda or not nv2.is_contiguous():
                    nv1, nv2 = nv1.to(device=dev, dtype=torch.int32).contiguous(), nv2.to(device=dev, dtype=torch.int32).contiguous()
            _lib.call('fgnn_pack_adjacency_pair', _lib.ptr(t1.contiguous()), _lib.ptr(t2.contiguous()),
                      _lib.ptr(nv1) if ragged else None, _lib.ptr(nv2) if ragged else None, B, nmax, N, _lib.ptr(st['bits']),
                      _lib.ptr(st['nv']) if ragged else None, _lib.ptr(st['inv']) if ragged else None, _lib.ptr(st['flag']),
                      _lib.stream_ptr())
            if first_of_shape or (self.INPUT_CHECK_EVERY and st['calls'] % self.INPUT_CHECK_EVERY == 0):
                self._raise_if_not_representation(st)
        elif ragged:
            if nmax < st['nmax']:                        # a smaller batch in the same workspace: the old values are padding now
                st['x'].zero_()
            st['x'][:B, :c0, :nmax, :nmax].copy_(t1)
            st['x'][B:, :c0, :nmax, :nmax].copy_(t2)
        else:
            st['x'][:B, :c0].copy_(t1)
            st['x'][B:, :c0].copy_(t2)
        st['nmax'] = nmax
        if st.get('flat') is not net._flat:             # (re)bound parameters: a captured graph holds the old addresses
            st['graph'], st['flat'] = {}, net._flat
        xin, bits = (None, st['bits']) if tr else (st['x'], None)

        def work():
            params = net._engine_params()                # the bound flat buffer, or its zero-padded image (an index_copy_)
            grads = net._flat_grad if pad is None else pad['pgrad']
            if ragged:
                # 1 / sum(n) on the device (one tiny launch): the scoring backward reads it as its gradient scale and the loss job of
                # the gradient-finalize launch multiplies it in -- no copy, no host round trip, nothing of the batch in the graph
                if not tr:      # (input_form='tensor_representation': the packing launch in front of the graph has written st['inv'])
                    _lib.call('fgnn_inv_node_count', _lib.ptr(st['nv']), B, _lib.ptr(st['inv']), _lib.stream_ptr())
                eng._loss_scale_dev = st['inv']
                scores, loss = eng.forward(params, xin, nvalid=st['nv'], total_nodes=1.0, defer_loss=True, bits=bits)
                eng.backward(params, grads, gscale_dev=st['inv'])
            else:
                scores, loss = eng.step(params, grads, xin, bits=bits)
            if pad is not None:
                torch.index_select(pad['pgrad'], 0, pad['idx'], out=net._flat_grad)
            if metric and builtin_metric:
                nvp = _lib.ptr(st['nv']) if ragged else None
                if self.metric is accuracy_max:
                    _lib.call('fgnn_accuracy_max', _lib.ptr(scores), nvp, B, N, _lib.ptr(st['correct']), _lib.stream_ptr())
                else:               # -log_softmax over the valid columns, then SciPy's assignment on the device (metrics.py)
                    s = scores
                    if ragged:
                        col = torch.arange(N, device=dev)[None, None, :] < st['nv'][:B, None, None]
                        s = s.masked_fill(~col, float('-inf'))
                    cost = (-torch.log_softmax(s, -1)).contiguous()
                    _lib.call('fgnn_lsap_accuracy', _lib.ptr(cost), N * N, N, nvp, B, N, _lib.ptr(st['correct']), None, _lib.stream_ptr())
            return scores, loss

        key = bool(metric and builtin_metric)
        if capture and st['graph'].get(key) is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # allocations, kernel attributes
                for _ in range(2):
                    work()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st['out', key] = work()
            st['graph'][key] = g
        if capture:
            st['graph'][key].replay()
            scores, loss = st['out', key]
        else:
            scores, loss = work()
        for p, v in zip(net._param_list, net._grad_views):
            if p.requires_grad:
                p.grad = v
        if ragged:
            scores = MaskedTensor(scores, x1.nvalid, (1, 2), x1.base_name)
        if not metric:
            return loss.reshape(()), scores
        if builtin_metric:
            total = st['nv'][:B].sum() if ragged else torch.tensor(B * N, device=dev)
            return loss.reshape(()), scores, (st['correct'].sum(), total)
        acc, n = self.metric(scores)
        return loss.reshape(()), scores, (acc, n)

    @staticmethod
    def _raise_if_not_representation(st):
        if int(st['flag'].item()) != 0:             # (one host synchronisation)
            st['flag'].zero_()
            raise RuntimeError("fused_step(input_form='tensor_representation'): a batch since the last check is NOT the tensor "
                               'representation of a 0/1 adjacency (channel 0 in {0, 1}, channel 1 = diag(row sums), '
                               'loaders/data_generator.py:118-125) -- the structured block 1 does not apply and the results of those '
                               "steps are invalid; run it through the dense path (input_form='dense')")

    def check_input_form(self):
        """input_form='tensor_representation': read the device verdict of every fused_step since the last check now (one host
        synchronisation); raises if one of those batches was not a tensor representation.  True if there was something to check."""
        flags = self.__dict__.get('_tr_flags', {})
        for flag in flags.values():
            self._raise_if_not_representation({'flag': flag})
        return bool(flags)

    def _captured_module_step(self, x1, x2, capture):
        """forward + loss + backward of the eager module path (models/trainers.py:60-76) as ONE replayed HIP graph per batch
        shape: the ~400 launches of a non-standard model cost the host nothing on replay.  Same contract as fused_step:
        every p.grad is overwritten per call, the returned tensors are overwritten by the next call of the same shape."""
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        params = [p for _, p in named]

        def run(a, b):
            # The step differentiates fresh leaves that alias the parameters' storage, not the parameters themselves: a
            # parameter's AccumulateGrad node lives as long as ANY autograd graph that reached it (e.g. the caller's previous
            # `loss`), belongs to the stream it was created on, and the autograd engine synchronises with that stream --
            # which is not allowed inside a capture (the capture then dies in hipStreamEndCapture).
            leaves = {n: p.detach().requires_grad_(True) for n, p in named}
            scores = torch.func.functional_call(self, leaves, (a, b))
            loss = self.loss(scores)
            grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
            return loss.detach().reshape(()), scores.detach(), grads

        if not capture:
            loss, scores, grads = run(x1, x2)
            for p, gr in zip(params, grads):
                p.grad = gr
            return loss, scores
        steps = self.__dict__.setdefault('_module_steps', {})
        key = (tuple(x1.shape), tuple(x2.shape), x1.dtype, x1.device, tuple(id(p) for p in params))
        st = steps.pop(key, None)
        if st is None:
            # a handful of batch shapes, least recently used first out (each holds a graph's private memory pool; a loader that
            # cycles through more shapes than MODULE_STEPS_MAX re-captures -- ~400 launches x 4 -- on every miss: raise the bound)
            while len(steps) >= max(1, int(self.MODULE_STEPS_MAX)):
                steps.pop(next(iter(steps)))
            st = {'x1': torch.empty_like(x1), 'x2': torch.empty_like(x2), 'graph': None}
        steps[key] = st                                 # (re-)inserted last = most recently used
        st['x1'].copy_(x1)
        st['x2'].copy_(x2)
        if st['graph'] is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # allocations, kernel attributes, autograd's first-use set-up
                for _ in range(3):
                    run(st['x1'], st['x2'])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, scores, grads = run(st['x1'], st['x2'])
            if all(gr is None for gr in grads):
                raise RuntimeError('fused_step: no parameter received a gradient from the captured module step')
            st['out'], st['grads'], st['graph'] = (loss, scores), grads, g
        st['graph'].replay()
        for p, gr in zip(params, st['grads']):
            p.grad = gr
        return st['out']

    # -- the step methods of the reference's LightningModule (models/trainers.py:70-90); `log` is a no-op
    #    here and is overridden by whatever training shell wraps the module
    def log(self, name, value, **kwargs):
        pass

    def _shared_step(self, batch, prefix):
        raw_scores = self(batch[0], batch[1])
        loss = self.loss(raw_scores)
        self.log(prefix + '_loss', loss)
        acc, n = self.metric(raw_scores)
        self.log(prefix + '_acc', acc / n)
        return loss

    def training_step(self, batch, batch_idx):
        return self._shared_step(batch, 'train')

    def validation_step(self, batch, batch_idx):
        self._shared_step(batch, 'val')

    def test_step(self, batch, batch_idx):
        self._shared_step(batch, 'test')

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.lr, amsgrad=False)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=self.scheduler_decay,
                                                               patience=self.scheduler_step, min_lr=self.lr_stop)
        return {'optimizer': optimizer, 'lr_scheduler': {'scheduler': scheduler, 'monitor': 'val_loss', 'frequency': 1}}
