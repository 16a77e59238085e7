"""Decision-pinned evaluation of the 2-FGNN hot path.  TEST INFRASTRUCTURE ONLY (same rules as fgnn_oracle.py: only
tests/ may import it; nothing under graph_neural_net_amd/ does).

A training step takes two kinds of DISCRETE decisions: the sign of every hidden pre-activation (`F.relu`,
models/layers.py:129-130) and the arg-max of every row of the pooling (`torch.max(x, -1)`, models/layers.py:202-203).  Between
two evaluations of the same step that take the same decisions the result is a smooth function of the arithmetic -- rounding
moves it by rounding; where they take ONE decision differently (a pre-activation within rounding distance of zero) the
gradients differ by 1e-4 ... 1e-2.  This module restates the reference's op sequence with the decisions as
INPUTS:

    relu(pre)            ->  where(mask, pre, 0)                 mask = [pre > 0] as SOME evaluation saw it
    max(x, -1)[0]        ->  gather(x, -1, idx)                  idx  = that evaluation's arg-max

so that an fp64 evaluation can follow the branch an fp32 engine took, and the engine's gradients can be held to a sharp
tolerance (1e-5 max-norm relative per tensor, tests/test_gpu_grad_pinned.py) instead of a distribution.

Parity status: PINNED.  Fed the reference's OWN decisions (forward hooks on the imported reference, tests/golden/make_golden.py
`round5`), every function below is `torch.equal` to the reference in fp32 and in fp64 -- scores, loss and all gradients
(tests/test_oracle_pinned.py live, tests/golden/pinned_decisions.npz as committed vectors).  Everything but the two substitutions
above is the op sequence of oracle/fgnn_oracle.py (each function cites the reference lines it follows there).

Device-agnostic pure PyTorch: the GPU tests run it in fp64 on the device (torch's own fp64 kernels, as the checker).
"""
import torch
import torch.nn.functional as F

from . import fgnn_oracle as O


def collect_decisions(x, sd):
    """The decisions of the plain oracle (= the reference) on the stacked batch x: ({(blk, mlp, layer): bool (G, C, N, N)}, idx
    (G, C, N) int64).  Used to pin this module against the reference and as the 'fp64 took these' side of the diagnostics."""
    masks = {}
    h = x
    for blk in range(1, O.num_blocks_of(sd) + 1):
        outs = {}
        for j in (1, 2):
            outs[j] = _mlp_collect(h, O.mlp_params(sd, blk, j), masks, blk, j)
        mult = torch.matmul(outs[1], outs[2])
        h = _mlp_collect(torch.cat((mult, h), dim=1), O.mlp_params(sd, blk, 3), masks, blk, 3)
    return masks, torch.max(h, -1)[1]


def _mlp_collect(x, p, masks, blk, j):
    ws, bs, gn_w, gn_b = p
    out = x
    for l, (w, b) in enumerate(zip(ws[:-1], bs[:-1])):
        pre = F.conv2d(out, w, b)
        masks[(blk, j, l)] = pre > 0
        out = F.relu(pre)
    return O.graph_norm(F.conv2d(out, ws[-1], bs[-1]), gn_w, gn_b)


def mlp_block_real_pinned(x, ws, bs, gn_w, gn_b, masks):
    """models/layers.py:126-131 with the ReLU decisions given: masks[l] is bool, shaped like the output of conv l."""
    out = x
    for l, (w, b) in enumerate(zip(ws[:-1], bs[:-1])):
        pre = F.conv2d(out, w, b)
        out = torch.where(masks[l], pre, torch.zeros((), dtype=pre.dtype, device=pre.device))
    return O.graph_norm(F.conv2d(out, ws[-1], bs[-1]), gn_w, gn_b)


def node_embedding_pinned(x, sd, masks, idx):
    """models/blocks_emb.py:16-43 with masks {(blk, mlp, layer): bool (G, C, N, N)} and the pooling's arg-max idx (G, C, N)."""
    h = x
    depth = len(O.mlp_params(sd, 1, 1)[0])
    for blk in range(1, O.num_blocks_of(sd) + 1):
        mk = lambda j: [masks[(blk, j, l)] for l in range(depth - 1)]
        m1 = mlp_block_real_pinned(h, *O.mlp_params(sd, blk, 1), mk(1))
        m2 = mlp_block_real_pinned(h, *O.mlp_params(sd, blk, 2), mk(2))
        mult = torch.matmul(m1, m2)
        h = mlp_block_real_pinned(torch.cat((mult, h), dim=1), *O.mlp_params(sd, blk, 3), mk(3))
    return torch.gather(h, -1, idx.unsqueeze(-1)).squeeze(-1)


def _loss_sum(scores):
    """toolbox/losses.py:27-34: sum of the per-graph cross entropies against target arange(n), and the node count."""
    loss, total = 0, 0
    for out in scores:
        n = out.shape[0]
        loss = loss + F.cross_entropy(out, torch.arange(n, device=out.device), reduction='sum')
        total += n
    return loss, total


def step_fwd_bwd_pinned(x1, x2, sd, masks, idx, dtype=torch.float64, device=None):
    """One step's model work on the branch given by the decisions.  x1, x2: (B, c0, N, N); masks / idx are those of the STACKED batch
    cat(x1, x2) (G = 2B graphs, the engine's order: first all left graphs, then all right graphs).  The two sides run as two
    forward passes like models/trainers.py:60-68.  Returns (scores, loss, grads{name: tensor}) in `dtype`."""
    device = x1.device if device is None else device
    params = {k: v.detach().to(device=device, dtype=dtype).clone().requires_grad_(True) for k, v in O._strip(sd).items()}
    B = x1.shape[0]
    side = lambda x, lo: node_embedding_pinned(x.to(device=device, dtype=dtype), params,
                                               {k: v[lo:lo + B].to(device) for k, v in masks.items()}, idx[lo:lo + B].to(device))
    e1, e2 = side(x1, 0), side(x2, B)
    scores = torch.matmul(torch.transpose(e1, 1, 2), e2)
    loss, total = _loss_sum(scores)
    loss = loss / total
    names = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in names])
    return scores.detach(), loss.detach(), dict(zip(names, gs))


def step_fwd_bwd_pinned_ragged(x1, x2, sizes, sd, masks, idx, dtype=torch.float64, device=None):
    """The ragged step (per-graph dense runs, the reference's own definition of a masked result:
    maskedtensors/test_maskedtensor.py:22-27, 141-150): x1, x2 (B, c0, Nmax, Nmax) zero-padded, sizes[b] vertices in pair b;
    decisions of the stacked padded batch, read on the valid corners.  loss = sum of the CE sums / sum(n) (toolbox/losses.py:27-34).
    Returns ([scores_b (n_b, n_b)], loss, grads)."""
    device = x1.device if device is None else device
    params = {k: v.detach().to(device=device, dtype=dtype).clone().requires_grad_(True) for k, v in O._strip(sd).items()}
    B = x1.shape[0]
    scores = []
    for b, n in enumerate(sizes):
        es = []
        for x, g in ((x1, b), (x2, B + b)):
            mk = {k: v[g:g + 1, :, :n, :n].to(device) for k, v in masks.items()}
            es.append(node_embedding_pinned(x[b:b + 1, :, :n, :n].to(device=device, dtype=dtype), params, mk, idx[g:g + 1, :, :n].to(device)))
        scores.append(torch.matmul(torch.transpose(es[0], 1, 2), es[1])[0])
    loss, total = _loss_sum(scores)
    loss = loss / total
    names = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in names])
    return [s.detach() for s in scores], loss.detach(), dict(zip(names, gs))


def pack_decisions(masks, idx):
    """-> {name: numpy array} for a fixture: masks bit-packed (numpy.packbits over the flattened tensor) + their shapes."""
    import numpy as np
    out = {'idx': idx.cpu().numpy().astype(np.int32)}
    for (blk, j, l), m in masks.items():
        out['mask/%d/%d/%d' % (blk, j, l)] = np.packbits(m.cpu().numpy().reshape(-1))
        out['shape/%d/%d/%d' % (blk, j, l)] = np.array(m.shape, dtype=np.int64)
    return out


def unpack_decisions(d, prefix=''):
    import numpy as np
    masks = {}
    for k in d:
        if k.startswith(prefix + 'mask/'):
            blk, j, l = [int(v) for v in k[len(prefix) + 5:].split('/')]
            shape = tuple(int(v) for v in np.asarray(d[prefix + 'shape/%d/%d/%d' % (blk, j, l)]))
            n = int(np.prod(shape))
            masks[(blk, j, l)] = torch.from_numpy(np.unpackbits(np.asarray(d[k]))[:n].reshape(shape).astype(bool))
    return masks, torch.from_numpy(np.asarray(d[prefix + 'idx']).astype(np.int64))
