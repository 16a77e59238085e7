#!/usr/bin/env python3
"""One MODEL configuration through the module surface against the oracle, per-tensor gradient errors (the zoom of gpu_fuzz_models.py).
usage (GPU box): python tests/diag/gpu_model_case.py c0 in out depth blocks B N [seed] [ragged sizes ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from graph_neural_net_amd.masked import from_list               # noqa: E402
from graph_neural_net_amd.siamese import Siamese_Node_Exp        # noqa: E402
from oracle import fgnn_oracle as O                              # noqa: E402
from util import rel                                             # noqa: E402

DEV = 'cuda:0'
c0, win, wout, depth, nblk, B, N = [int(v) for v in sys.argv[1:8]]
seed = int(sys.argv[8]) if len(sys.argv) > 8 else 0
ns = [int(v) for v in sys.argv[9:]] or [N] * B
ragged = len(sys.argv) > 9
torch.manual_seed(seed)
sd = O.init_state_dict(original_features_num=c0, num_blocks=nblk, in_features=win, out_features=wout, depth_of_mlp=depth)
g = torch.Generator().manual_seed(seed)
for k in sd:
    if k.endswith('.bias') and sd[k].dim() == 1:
        sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    elif k.endswith('gn.weight'):
        sd[k] = sd[k] * (1.0 + 0.2 * torch.randn(sd[k].shape, generator=g))
    elif k.endswith('gn.bias'):
        sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
xs = [torch.randn(c0, n, n, generator=g) for n in ns]
ys = [torch.randn(c0, n, n, generator=g) for n in ns]
s32, l32, g32 = O.step_fwd_bwd_ragged(xs, ys, sd)
s64, l64, g64 = O.step_fwd_bwd_ragged([t.double() for t in xs], [t.double() for t in ys], {k: v.double() for k, v in sd.items()})
ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=nblk, in_features=win, out_features=wout,
          depth_of_mlp=depth, constant_n_vertices=not ragged)
model = Siamese_Node_Exp(c0, ne).to(DEV)
model.load_state_dict({'node_embedder.' + k: v for k, v in sd.items()})
net = model.node_embedder
print('path:', 'conv.hip modules' if net._standard_layout() is None else ('padded engine' if net._pad is not None else 'engine'))
if ragged:
    scores = model(from_list([t.to(DEV) for t in xs], dims=(1, 2), base_name='N'), from_list([t.to(DEV) for t in ys], dims=(1, 2), base_name='M'))
else:
    scores = model(torch.stack(xs).to(DEV), torch.stack(ys).to(DEV))
loss = model.loss(scores)
loss.backward()
torch.cuda.synchronize()
for i, (a, b32, b64) in enumerate(zip(list(scores), s32, s64)):
    print('scores pair %d: ours %.2e oracle32 %.2e' % (i, rel(a.detach().cpu(), b64), rel(b32, b64)))
print('loss', loss.item(), l64.item())
for n, p in model.named_parameters():
    k = n[len('node_embedder.'):]
    print('%-40s ours %.2e  oracle32 %.2e  |g| %.2e' % (k, rel(p.grad.cpu(), g64[k]), rel(g32[k], g64[k]), g64[k].abs().max().item()))
