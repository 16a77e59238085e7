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
    """Whole node-embedder forward/backward through FgnnEngine (one autograd node)."""

    @staticmethod
    def forward(ctx, net, x, nvalid, flat):
        eng = net._engine_for(x, nvalid)
        eng.embed(flat, x, nvalid)
        ctx.net, ctx.eng = net, eng
        ctx.save_for_backward(flat)
        return eng.E.clone()

    @staticmethod
    def backward(ctx, dE):
        (flat,) = ctx.saved_tensors
        grads = torch.zeros_like(flat)
        ctx.eng.backward_from_dE(flat, grads, dE.contiguous())
        return None, None, None, grads


class Network(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.graph = build_graph(net)
        for path, (node, _) in self.graph.items():
            setattr(self, path.replace(SEP, '_'), node)
        self._layout = None
        self._engines = {}

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

    def _flat_params(self):
        lay = self._layout
        ps = [p for _, p in self.named_parameters()]
        return torch.cat([p.reshape(-1) for p in ps])

    def _engine_for(self, x, nvalid):
        key = (x.shape[0], x.shape[-1], nvalid is not None, x.device)
        if key not in self._engines:
            self._engines[key] = FgnnEngine(self._layout, x.shape[0], x.shape[-1], x.device, ragged=nvalid is not None)
        return self._engines[key]

    def fused_embedding(self, x):
        """x: (G, c0, N, N) tensor or MaskedTensor -> node embeddings (G, C, N)."""
        lay = self._standard_layout()
        if lay is None:
            return self.forward({'input': x})['ne/suffix']
        t, nvalid = (x.tensor, x.nvalid) if isinstance(x, MaskedTensor) else (x, None)
        if not t.is_cuda:
            raise RuntimeError('graph_neural_net_amd only runs on the GPU (input is on %s)' % (t.device,))
        flat = self._flat_params()          # differentiable concat: grads flow back to every Parameter
        e = _EmbedFn.apply(self, t.contiguous(), nvalid, flat)
        if isinstance(x, MaskedTensor):
            return MaskedTensor(e, nvalid, (2,), x.base_name)
        return e
