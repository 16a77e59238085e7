"""Dict-graph executor: the reference's ``Network`` (models/utils.py:45-69).

A nested dict of nodes is flattened into path-keyed entries (``'ne/bm/block1/mlp1'``);
a node without explicit inputs reads the previous entry, relative input names are
resolved against the node's own sub-graph.  Modules are registered under
``path.replace('/', '_')``, which fixes the ``state_dict`` key names
(``ne_bm_block1_mlp1.convs.0.weight`` ...).  ``forward`` returns every node output.

``Network.fused_embedding`` is the fast path used by ``Siamese_Node_Exp``: when the graph
is the standard ``node_embedding`` stack it runs the whole embedder forward (and, through
autograd, backward) as the fused kernel sequence of ``engine.FgnnEngine`` and returns only
``'ne/suffix'``.
"""
import torch
import torch.nn as nn

from . import _lib
from .engine import EngineCache, FgnnEngine, ParamLayout
from .masked import MaskedTensor

SEP = '/'


def _flatten(net, prefix=()):
    for name, node in net.items():
        if isinstance(node, dict):
            yield from _flatten(node, prefix + (name,))
        else:
            yield prefix + (name,), node


def _resolve(parts):
    out = []
    for p in parts:
        if p == '..':
            out.pop()
        else:
            out.append(p)
    return SEP.join(out)


def build_graph(net):
    """{path: (node, [absolute input paths])} in execution (insertion) order."""
    flat = [(SEP.join(path), node) for path, node in _flatten(net)]
    graph = {}
    for i, (path, node) in enumerate(flat):
        fn, ins = node if type(node) is tuple else (node, [-1])
        parent = path.split(SEP)[:-1]
        resolved = []
        for ref in ins:
            if isinstance(ref, str):
                resolved.append(_resolve(parent + ref.split(SEP)))
            else:
                resolved.append(flat[i + ref][0])
        graph[path] = (fn, resolved)
    return graph


def _embed_forward(ctx, net, x, nvalid):
    params, xin = net._engine_params(), net._pad_input(x, nvalid, True)
    eng = net._engine_for(xin, nvalid, True)
    eng.embed(params, xin, nvalid)
    eng.generation = getattr(eng, 'generation', 0) + 1
    ctx.net, ctx.eng, ctx.generation, ctx.xshape = net, eng, eng.generation, tuple(x.shape)
    return net._crop_embedding(eng.E)


def _embed_backward(ctx, dE, target, want_dx):
    """Engine backward into the flat gradient buffer `target`; returns the input gradient (or None)."""
    net, eng = ctx.net, ctx.eng
    if eng.generation != ctx.generation:
        raise RuntimeError('Network: the activations of this forward pass were overwritten by a later forward of the '
                           'same shape before backward() ran (the fused engine keeps one workspace per shape); call '
                           'backward() before the next forward, or run the other forward under torch.no_grad()')
    dx = None
    if want_dx:
        if getattr(net, 'precision', 'fp32') == 'bf16':
            raise RuntimeError('Network: the gradient with respect to the input is only built for the fp32 engine')
        dx = torch.zeros(eng.G, eng.layout.c0, eng.N, eng.N, dtype=torch.float32, device=dE.device)
    dE = net._pad_dE(dE.contiguous())
    if net._pad is None:
        eng.backward_from_dE(net._flat, target, dE, **({'dx': dx} if dx is not None else {}))
    else:                                              # padded engine: gather the real entries of its gradient
        eng.backward_from_dE(net._pad['pflat'], net._pad['pgrad'], dE, **({'dx': dx} if dx is not None else {}))
        torch.index_select(net._pad['pgrad'], 0, net._pad['idx'], out=target)
        if dx is not None:
            dx = dx[:, :ctx.xshape[1]].contiguous()
    return dx


class _Outputs(dict):
    """Network.forward's result: every node output by path.  A Concat output that only fused MLP kernels read is kept as its parts
    (layers.LazyCat) and becomes the concatenated tensor when somebody looks at it."""

    def raw(self, key):
        return dict.__getitem__(self, key)

    def _out(self, key):
        v = dict.__getitem__(self, key)
        return v.materialize() if hasattr(v, 'materialize') else v

    def __getitem__(self, key):
        return self._out(key)

    def get(self, key, default=None):
        return self._out(key) if key in self else default

    def items(self):
        return [(k, self._out(k)) for k in self.keys()]

    def values(self):
        return [self._out(k) for k in self.keys()]

    def __iter__(self):                     # (also keeps dict(outputs) / {**outputs} off the C fast path that copies the raw values)
        return iter(list(dict.keys(self)))

    def copy(self):
        return dict(self.items())


class _EmbedFn(torch.autograd.Function):
    """Whole node-embedder forward/backward through FgnnEngine (one autograd node) -- the fast mode.

    The module's 96 parameters are views into ONE persistent flat buffer (Network._bind_flat), so the engine reads
    them in place; their gradients land in ONE persistent flat gradient buffer whose per-parameter views are
    attached as ``p.grad`` -- no per-step torch.cat / split, no 96 AccumulateGrad nodes.  The single differentiable
    input `anchor` only makes this node part of the graph (the input x takes part when it requires grad).
    Parameters are NOT autograd inputs here: no AccumulateGrad node, no parameter hook fires -- Network uses
    _EmbedFnParams instead whenever that matters (torch.distributed initialised, hooks registered, param_autograd=True)."""

    @staticmethod
    def forward(ctx, net, x, nvalid, anchor):
        return _embed_forward(ctx, net, x, nvalid)

    @staticmethod
    def backward(ctx, dE):
        net = ctx.net
        params = net._param_list
        views = net._grad_views
        live = [p.requires_grad for p in params]           # frozen parameters get no .grad
        fresh = all(p.grad is None for p in params)        # the usual case after zero_grad(set_to_none=True)
        target = net._flat_grad if fresh else net._flat_grad_tmp
        dx = _embed_backward(ctx, dE, target, ctx.needs_input_grad[1])
        if fresh:
            for p, v, on in zip(params, views, live):
                if on:
                    p.grad = v
        elif all((p.grad is v) or not on for p, v, on in zip(params, views, live)):
            net._flat_grad.add_(target)
        else:                                              # accumulate into whatever .grad holds
            for p, (off, n, shape), on in zip(params, net._param_spans, live):
                if not on:
                    continue
                g = target[off:off + n].view(shape)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.add_(g)
        return None, dx, None, torch.zeros_like(net._anchor)


class _EmbedFnParams(torch.autograd.Function):
    """The same engine calls with every parameter as an autograd input: one gradient per parameter is returned, so
    AccumulateGrad nodes, parameter hooks, DistributedDataParallel's reducer and torch.autograd.grad(loss, [p]) all work as
    with the reference's plain autograd (models/trainers.py:60-76) -- at the price of ~100 small host-side tensor
    operations per step."""

    @staticmethod
    def forward(ctx, net, x, nvalid, *params):
        return _embed_forward(ctx, net, x, nvalid)

    @staticmethod
    def backward(ctx, dE):
        net = ctx.net
        target = net._flat_grad_tmp
        dx = _embed_backward(ctx, dE, target, ctx.needs_input_grad[1])
        grads = tuple(target[off:off + n].view(shape).clone() if need else None
                      for (off, n, shape), need in zip(net._param_spans, ctx.needs_input_grad[3:]))
        return (None, dx, None) + grads


class Network(nn.Module):
    ENGINE_CACHE_BYTES = 8 << 30      # workspace budget of the per-shape engine cache (LRU, engine.EngineCache)

    def __init__(self, net):
        super().__init__()
        self.graph = build_graph(net)
        for path, (node, _) in self.graph.items():
            setattr(self, path.replace(SEP, '_'), node)
        self._layout = None
        self._lazy = None         # Concat nodes read by fused MLP kernels only (_lazy_cat_nodes)
        self._fan = set()         # outputs whose readers share one gradient buffer (layers.fan_out)
        self._pad = None          # widths below 32 embedded in the 32-wide engine by zero padding (_padded_layout)
        self._engines = EngineCache(self.ENGINE_CACHE_BYTES)
        self._flat = None
        # None: parameters become autograd inputs of the fused node only when something needs it (torch.distributed is
        # initialised with more than one rank, a parameter carries hooks); True / False force the mode (_EmbedFnParams)
        self.param_autograd = None

    def nodes(self):
        return (node for node, _ in self.graph.values())

    def _lazy_cat_nodes(self):
        """Concat nodes whose every reader is an MlpBlock_Real: their output stays a LazyCat (the parts) unless someone asks for the
        tensor -- the block's mlp3 reads [mult ; in] in place (csrc/mlp64.hip) instead of a concatenated copy."""
        if self._lazy is None:
            from .layers import Concat, MlpBlock_Real
            readers = {}
            for path, (node, ins) in self.graph.items():
                for name in ins:
                    readers.setdefault(name, []).append(node)
            self._lazy = {path for path, (node, _) in self.graph.items()
                          if isinstance(node, Concat) and readers.get(path) and all(isinstance(r, MlpBlock_Real) for r in readers[path])}
            # outputs read by two or more (fused-capable) MLPs, directly or through such a Concat: their gradient is summed inside the
            # MLP backward kernels (layers.fan_out) instead of by autograd's add kernels
            n64 = {}
            for path, (node, ins) in self.graph.items():
                if (isinstance(node, MlpBlock_Real) and not node.fused()) or path in self._lazy:
                    for name in ins:
                        n64[name] = n64.get(name, 0) + 1
            self._fan = {name for name, n in n64.items() if n >= 2}
        return self._lazy

    def forward(self, inputs):
        from .layers import LazyCat, MlpBlock_Real, fan_out, prepack64
        lazy = self._lazy_cat_nodes()
        mlps = [node for node, _ in self.graph.values() if isinstance(node, MlpBlock_Real)]
        prepack64(mlps)
        outputs = _Outputs(inputs)
        try:
            for path, (node, ins) in self.graph.items():
                if path not in outputs:
                    args = [outputs.raw(name) for name in ins]
                    if path in lazy:
                        outputs[path] = LazyCat(args)
                    else:
                        if not isinstance(node, MlpBlock_Real):
                            args = [a.materialize() if isinstance(a, LazyCat) else a for a in args]
                        outputs[path] = node(*args)
                    if path in self._fan:
                        outputs[path] = fan_out(outputs.raw(path))
        finally:
            for m in mlps:                  # an operand record is good for THIS pass only (a node that did not run must not keep one)
                m._packed64 = None
        return outputs

    # ------------------------------------------------------------------ fused fast path
    def _standard_layout(self):
        """ParamLayout if this graph is {'input', 'ne': node_embedding(...)}, else None."""
        if self._layout is not None:
            return self._layout or None
        from .layers import MlpBlock_Real
        try:
            k = 0
            while hasattr(self, 'ne_bm_block%d_mlp1' % (k + 1)):
                k += 1
            m1 = getattr(self, 'ne_bm_block1_mlp1')
            assert k > 0 and isinstance(m1, MlpBlock_Real) and 'ne/suffix' in self.graph
            c0 = m1.convs[0].in_channels
            depth = len(m1.convs)
            last = getattr(self, 'ne_bm_block%d_mlp3' % k)
            lay = ParamLayout(c0, k, m1.convs[0].out_channels if k > 1 else 32, last.convs[-1].out_channels, depth)
            named = list(self.named_parameters())
            assert [n for n, _ in named] == [e[0] for e in lay.entries]
            # every tensor must have the engine's shape (names alone do not tell a 32 -> 16 model from the 32 -> 32 one)
            assert all(tuple(p.shape) == tuple(e[2]) for (_, p), e in zip(named, lay.entries))
            self._layout = lay
        except (AssertionError, AttributeError, RuntimeError):
            self._layout = self._padded_layout() or False
        return self._layout or None

    def _padded_layout(self):
        """Channel widths below the engine's (original_features_num <= 32, in_features / out_features <= 32): the same graph is
        run by the 32-wide fused engine on zero-padded parameters and inputs -- the extra channels carry exact zeros through
        conv / ReLU / GraphNorm (weight 0, bias 0) / matmul / pooling, so the results are those of the narrow model.  Returns the
        padded ParamLayout and fills self._pad = {idx: position of every real parameter element inside the padded flat buffer,
        ...}, or None when the graph does not fit."""
        from .layers import MlpBlock_Real
        try:
            k = 0
            while hasattr(self, 'ne_bm_block%d_mlp1' % (k + 1)):
                k += 1
            assert k > 0 and 'ne/suffix' in self.graph
            blocks = [[getattr(self, 'ne_bm_block%d_mlp%d' % (b, j)) for j in (1, 2, 3)] for b in range(1, k + 1)]
            depth = len(blocks[0][0].convs)
            c0 = blocks[0][0].convs[0].in_channels
            assert 1 <= depth <= _lib.FGNN_MAX_DEPTH and c0 <= 32
            widths, last = [], c0
            for m1, m2, m3 in blocks:
                w = m1.convs[-1].out_channels
                assert w <= 32
                for m, cin in ((m1, last), (m2, last), (m3, w + last)):
                    assert isinstance(m, MlpBlock_Real) and len(m.convs) == depth and m.convs[0].in_channels == cin
                    assert all(c.out_channels == w and c.bias is not None for c in m.convs) and m.gn.features[1] == w
                    assert m.gn.weight is not None
                widths.append(w)
                last = w
            c0p = 2 if c0 <= 2 else 32
            lay = ParamLayout(c0p, k, 32, 32, depth)
            named = list(self.named_parameters())
            assert [n for n, _ in named] == [e[0] for e in lay.entries]
            idx = []
            for (name, off, shape), (_, p) in zip(lay.entries, named):
                b = int(name.split('block')[1].split('_')[0])
                w, lin = widths[b - 1], (c0 if b == 1 else widths[b - 2])
                if p.dim() == 4 and p.shape[0] != 1:             # conv weight (o, ci, 1, 1) inside (32, cp, 1, 1)
                    o, ci, cp = p.shape[0], p.shape[1], shape[1]
                    cols = torch.arange(ci)
                    if '_mlp3.convs.0.' in name:                  # [mult (w) ; block input (lin)] -> [32 ; padded input]
                        cols = torch.where(cols < w, cols, 32 + cols - w)
                    idx.append((off + torch.arange(o)[:, None] * cp + cols[None, :]).reshape(-1))
                else:                                             # conv bias (o,), gn weight / bias (1, o, 1, 1)
                    idx.append(off + torch.arange(p.numel()))
            self._pad = {'idx': torch.cat(idx), 'c0': c0, 'c0p': c0p, 'cout': widths[-1], 'total': lay.total}
            return lay
        except (AssertionError, AttributeError, RuntimeError, IndexError, ValueError):
            self._pad = None
            return None

    def _engine_params(self):
        """The flat parameter buffer the engine reads: the bound one, or its zero-padded image."""
        if self._pad is None:
            return self._flat
        P = self._pad
        if P.get('pflat') is None or P['pflat'].device != self._flat.device:
            P['idx'] = P['idx'].to(self._flat.device)
            P['pflat'] = torch.zeros(P['total'], dtype=torch.float32, device=self._flat.device)
            P['pgrad'] = torch.zeros_like(P['pflat'])
        P['pflat'].index_copy_(0, P['idx'], self._flat)
        return P['pflat']

    def _pad_input(self, x, nvalid, grad):
        """Input widths 3..31: the zero-padded 32-channel staging buffer belongs to the ENGINE that will read it (the
        fp32 engine keeps it by reference for block 1's backward, so the grad and the no-grad engine of one shape must not
        share it: an evaluation forward between a training forward and its backward would overwrite the saved input)."""
        P = self._pad
        if P is None or P['c0p'] == P['c0']:
            return x
        shape = (x.shape[0], P['c0p'], x.shape[2], x.shape[3])
        eng = self._engine_for_shape(x.shape[0], x.shape[-1], nvalid is not None, x.device, grad)
        buf = getattr(eng, '_xpad', None)
        if buf is None:
            buf = eng._xpad = torch.zeros(shape, dtype=torch.float32, device=x.device)
        buf[:, :P['c0']].copy_(x)
        return buf

    def _crop_embedding(self, E):
        if self._pad is None or self._pad['cout'] == E.shape[1]:
            return E.clone()
        return E[:, :self._pad['cout']].contiguous()

    def _pad_dE(self, dE):
        if self._pad is None or self._pad['cout'] == 32:
            return dE
        full = torch.zeros(dE.shape[0], 32, dE.shape[2], dtype=dE.dtype, device=dE.device)
        full[:, :dE.shape[1]].copy_(dE)
        return full

    def _bind_flat(self):
        """Move every parameter of the standard layout into one flat fp32 buffer (the parameters become views of it, in
        the reference's named_parameters() order) and prepare the flat gradient buffer + its per-parameter views.
        Re-done when a parameter was re-assigned or moved to another device."""
        if self._flat is not None:
            # per-forward check against the LIVE parameters (a re-assigned nn.Parameter, load_state_dict(assign=True), a
            # .data swap or .to(device) all re-bind): identity of every slot and the storage offset of every view
            base = self._flat.data_ptr()
            if all(m._parameters.get(n) is p and p.data_ptr() == base + 4 * off
                   for (m, n), p, (off, _, _) in zip(self._param_slots, self._param_list, self._param_spans)) \
                    and self._param_list[0].device == self._flat.device:
                return
        slots = [(m, n) for m in self.modules() for n, p in m._parameters.items() if p is not None]
        ps = [m._parameters[n] for m, n in slots]
        assert [id(p) for p in ps] == [id(p) for _, p in self.named_parameters()]
        self._param_slots = slots
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        spans, off = [], 0
        for p in ps:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            spans.append((off, n, tuple(p.shape)))
            off += n
        self._flat, self._param_list, self._param_spans = flat, ps, spans
        self._flat_grad = torch.zeros_like(flat)
        self._flat_grad_tmp = torch.empty_like(flat)
        self._grad_views = [self._flat_grad[o:o + n].view(shape) for o, n, shape in spans]
        self._anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)

    def _engine_for(self, x, nvalid, grad):
        return self._engine_for_shape(x.shape[0], x.shape[-1], nvalid is not None, x.device, grad)

    def _engine_for_shape(self, G, N, ragged, device, grad, block1=None):
        # separate workspaces for grad / no-grad forwards: an evaluation forward between a training forward and its
        # backward must not overwrite the saved activations.  `grad` is passed in by the caller: inside an
        # autograd.Function's forward torch.is_grad_enabled() is always False.
        # The cache is bounded (LRU by bytes, ENGINE_CACHE_BYTES): a stream of ragged shapes re-uses a few workspaces; an
        # engine evicted between a forward and its backward stays alive through the autograd node that holds it.
        bf16 = getattr(self, 'precision', 'fp32') == 'bf16'
        # 'step': Siamese_Node_Exp.fused_step; block1='structured': its tensor-representation form (bit-packed input, csrc/block1_struct.hip)
        key = (G, N, ragged, device, grad if isinstance(grad, str) else bool(grad), bf16) + ((block1,) if block1 else ())

        def make():
            if bf16:
                from .engine16 import FgnnEngineBF16
                return FgnnEngineBF16(self._layout, G, N, device, ragged=ragged, block1=block1)
            return FgnnEngine(self._layout, G, N, device, ragged=ragged, mfma='f32', block1=block1)
        self._engines.budget = self.ENGINE_CACHE_BYTES
        nbytes = EngineCache.engine_bytes(G, N, self._layout.num_blocks, 2 if bf16 else 4) // (1 if grad else 2)
        return self._engines.get(key, make, nbytes)[0]

    def _params_as_inputs(self):
        """Does this forward need per-parameter autograd (AccumulateGrad nodes, hooks, DDP's reducer)?"""
        if self.param_autograd is not None:
            return bool(self.param_autograd)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return True
        return any(p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None) for p in self._param_list)

    def half(self):
        """models/utils.py:71-74: the reference's 16-bit switch for the dict-graph.  The HIP kernels keep fp32 master
        parameters in every precision mode; 16-bit execution is the bf16 engine (engine16.FgnnEngineBF16), selected
        with Network.precision = 'bf16'.  half() selects it and returns self, like the reference's helper."""
        self.precision = 'bf16'
        self._engines.clear()
        return self

    def fused_embedding(self, x):
        """x: (G, c0, N, N) tensor or MaskedTensor -> node embeddings (G, C, N)."""
        lay = self._standard_layout()
        if lay is None:
            if getattr(self, 'precision', 'fp32') == 'bf16':
                raise RuntimeError('Network: the bf16 path is the fused engine (original_features_num = 2, in_features = '
                                   'out_features = 32); this graph runs through the per-layer fp32 modules')
            return self.forward({'input': x})['ne/suffix']
        t, nvalid = (x.tensor, x.nvalid) if isinstance(x, MaskedTensor) else (x, None)
        if not t.is_cuda:
            raise RuntimeError('graph_neural_net_amd only runs on the GPU (input is on %s)' % (t.device,))
        self._bind_flat()
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self._param_list) or t.requires_grad):
            if self._params_as_inputs():
                e = _EmbedFnParams.apply(self, t.contiguous(), nvalid, *self._param_list)
            else:
                e = _EmbedFn.apply(self, t.contiguous(), nvalid, self._anchor)
        else:
            xin = self._pad_input(t.contiguous(), nvalid, False)
            eng = self._engine_for(xin, nvalid, False)
            eng.embed(self._engine_params(), xin, nvalid)
            e = self._crop_embedding(eng.E)
        if isinstance(x, MaskedTensor):
            return MaskedTensor(e, nvalid, (2,), x.base_name)
        return e
