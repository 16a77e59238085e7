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
from .engine import FgnnEngine, ParamLayout
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


class _EmbedFn(torch.autograd.Function):
    """Whole node-embedder forward/backward through FgnnEngine (one autograd node).

    The module's 96 parameters are views into ONE persistent flat buffer (Network._bind_flat), so the engine reads
    them in place; their gradients land in ONE persistent flat gradient buffer whose per-parameter views are
    attached as ``p.grad`` -- no per-step torch.cat / split, no 96 AccumulateGrad nodes.  The single differentiable
    input `anchor` only makes this node part of the graph."""

    @staticmethod
    def forward(ctx, net, x, nvalid, anchor):
        eng = net._engine_for(x, nvalid, True)
        eng.embed(net._flat, x, nvalid)
        eng.generation = getattr(eng, 'generation', 0) + 1
        ctx.net, ctx.eng, ctx.generation = net, eng, eng.generation
        return eng.E.clone()

    @staticmethod
    def backward(ctx, dE):
        net, eng = ctx.net, ctx.eng
        if eng.generation != ctx.generation:
            raise RuntimeError('Network: the activations of this forward pass were overwritten by a later forward of the '
                               'same shape before backward() ran (the fused engine keeps one workspace per shape); call '
                               'backward() before the next forward, or run the other forward under torch.no_grad()')
        params = net._param_list
        views = net._grad_views
        if all(p.grad is None for p in params):            # the usual case after zero_grad(set_to_none=True)
            eng.backward_from_dE(net._flat, net._flat_grad, dE.contiguous())
            for p, v in zip(params, views):
                p.grad = v
        else:                                              # accumulate into whatever .grad holds
            tmp = net._flat_grad_tmp
            eng.backward_from_dE(net._flat, tmp, dE.contiguous())
            if all(p.grad is v for p, v in zip(params, views)):
                net._flat_grad.add_(tmp)
            else:
                for p, (off, n, shape) in zip(params, net._param_spans):
                    g = tmp[off:off + n].view(shape)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.add_(g)
        return None, None, None, torch.zeros_like(net._anchor)


class Network(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.graph = build_graph(net)
        for path, (node, _) in self.graph.items():
            setattr(self, path.replace(SEP, '_'), node)
        self._layout = None
        self._engines = {}
        self._flat = None

    def nodes(self):
        return (node for node, _ in self.graph.values())

    def forward(self, inputs):
        outputs = dict(inputs)
        for path, (node, ins) in self.graph.items():
            if path not in outputs:
                outputs[path] = node(*[outputs[name] for name in ins])
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
            lay = ParamLayout(c0, k, m1.convs[0].out_channels, m1.convs[-1].out_channels, depth)
            names = [n for n, _ in self.named_parameters()]
            assert names == [e[0] for e in lay.entries]
            self._layout = lay
        except (AssertionError, AttributeError, RuntimeError):
            self._layout = False
        return self._layout or None

    def _bind_flat(self):
        """Move every parameter of the standard layout into one flat fp32 buffer (the parameters become views of it, in
        the reference's named_parameters() order) and prepare the flat gradient buffer + its per-parameter views.
        Re-done when a parameter was re-assigned or moved to another device."""
        if self._flat is not None:
            # cheap per-forward check (first and last parameter still live inside the flat buffer); a full walk over the
            # 96 parameters costs more host time than a kernel launch
            first, last = self._param_list[0], self._param_list[-1]
            if (first.data_ptr() == self._flat.data_ptr() and first.device == self._flat.device
                    and last.data_ptr() == self._flat.data_ptr() + 4 * self._param_spans[-1][0]):
                return
        ps = [p for _, p in self.named_parameters()]
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
        # separate workspaces for grad / no-grad forwards: an evaluation forward between a training forward and its
        # backward must not overwrite the saved activations.  `grad` is passed in by the caller: inside an
        # autograd.Function's forward torch.is_grad_enabled() is always False.
        key = (x.shape[0], x.shape[-1], nvalid is not None, x.device, bool(grad))
        if key not in self._engines:
            if getattr(self, 'precision', 'fp32') == 'bf16':
                from .engine16 import FgnnEngineBF16
                self._engines[key] = FgnnEngineBF16(self._layout, x.shape[0], x.shape[-1], x.device, ragged=nvalid is not None)
            else:
                self._engines[key] = FgnnEngine(self._layout, x.shape[0], x.shape[-1], x.device, ragged=nvalid is not None)
        return self._engines[key]

    def half(self):
        """models/utils.py:71-74: the reference's 16-bit switch for the dict-graph.  The HIP kernels keep fp32 master
        parameters in every precision mode; 16-bit execution is the bf16 engine (engine16.FgnnEngineBF16), selected
        with Network.precision = 'bf16'.  half() selects it and returns self, like the reference's helper."""
        self.precision = 'bf16'
        self._engines = {}
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
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list):
            e = _EmbedFn.apply(self, t.contiguous(), nvalid, self._anchor)
        else:
            eng = self._engine_for(t, nvalid, False)
            eng.embed(self._flat, t.contiguous(), nvalid)
            e = eng.E.clone()
        if isinstance(x, MaskedTensor):
            return MaskedTensor(e, nvalid, (2,), x.base_name)
        return e
