"""CPU oracle for the 2-FGNN hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``graph_neural_net_amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back to this code.

It is a functional, pure-PyTorch-CPU restatement of the reference's algorithm
(the reference is 100 % Python on ATen; the parity target is therefore "ATen
CPU fp32 of this torch build").  It issues the same ATen
op sequence as the reference so that, in the build container where
``/root/reference`` can be imported, it is ``torch.equal`` to the reference
(checked by ``tests/golden/make_golden.py`` and
``tests/test_oracle_pinned.py``).  On the GPU box it is pinned by the golden
vectors committed under ``tests/golden/`` that were produced by the reference
itself.

Parity status: PINNED (golden vectors generated from the imported reference,
fp32 and fp64; see tests/golden/README.md).

Reference lines each function follows (paths under /root/reference):
  normalize            models/layers.py:71-80
  graph_norm           models/layers.py:47-69
  mlp_block_real       models/layers.py:109-131
  fgnn_block           models/blocks_emb.py:16-27, models/layers.py:145-146,161-162
  node_embedding       models/blocks_emb.py:29-43, models/layers.py:194-203
  siamese_scores       models/trainers.py:60-68
  triplet_loss_mean    toolbox/losses.py:8-34 ('mean');  triplet_loss_mean_of_mean: the 'mean_of_mean' reduction (:14-15)
  pad_graph_list       maskedtensors/maskedtensor.py:8-48
"""
import math

import torch
import torch.nn.functional as F

EPS = 1e-05


# ----------------------------------------------------------------------------
# parameter access: the reference's state_dict key layout (models/utils.py:57-58)
# ----------------------------------------------------------------------------
def _strip(sd):
    """Accept both `node_embedder.ne_bm_...` and `ne_bm_...` key styles."""
    out = {}
    for k, v in sd.items():
        if k.startswith('node_embedder.'):
            k = k[len('node_embedder.'):]
        out[k] = v
    return out


def mlp_params(sd, blk, j):
    """Return ([W_i], [b_i], gn_w, gn_b) of `ne_bm_block{blk}_mlp{j}`."""
    sd = _strip(sd)
    pfx = 'ne_bm_block%d_mlp%d.' % (blk, j)
    ws, bs = [], []
    i = 0
    while pfx + 'convs.%d.weight' % i in sd:
        ws.append(sd[pfx + 'convs.%d.weight' % i])
        bs.append(sd[pfx + 'convs.%d.bias' % i])
        i += 1
    return ws, bs, sd[pfx + 'gn.weight'], sd[pfx + 'gn.bias']


def num_blocks_of(sd):
    sd = _strip(sd)
    k = 0
    while 'ne_bm_block%d_mlp1.convs.0.weight' % (k + 1) in sd:
        k += 1
    return k


# ----------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------
def normalize(b, eps=EPS):
    """models/layers.py:71-80 with constant_n_vertices=True."""
    means = torch.mean(b, dim=(-1, -2), keepdim=True)
    vars = torch.var(b, unbiased=False, dim=(-1, -2), keepdim=True)
    n = b.size(-1)
    return (b - means) / (2 * torch.sqrt(n * (vars + eps)))


def graph_norm(b, weight, bias, eps=EPS):
    """models/layers.py:68-69; weight/bias are (1,C,1,1)."""
    return weight * normalize(b, eps=eps) + bias


def mlp_block_real(x, ws, bs, gn_w, gn_b, eps=EPS):
    """models/layers.py:126-131: conv1x1+relu ... last conv without relu, GraphNorm."""
    out = x
    for w, b in zip(ws[:-1], bs[:-1]):
        out = F.relu(F.conv2d(out, w, b))
    return graph_norm(F.conv2d(out, ws[-1], bs[-1]), gn_w, gn_b, eps=eps)


def fgnn_block(x, sd, blk, keep=None):
    """models/blocks_emb.py:16-27.  Returns mlp3 output; fills `keep` with intermediates."""
    m1 = mlp_block_real(x, *mlp_params(sd, blk, 1))
    m2 = mlp_block_real(x, *mlp_params(sd, blk, 2))
    mult = torch.matmul(m1, m2)
    cat = torch.cat((mult, x), dim=1)
    m3 = mlp_block_real(cat, *mlp_params(sd, blk, 3))
    if keep is not None:
        pfx = 'ne/bm/block%d/' % blk
        keep[pfx + 'mlp1'] = m1
        keep[pfx + 'mlp2'] = m2
        keep[pfx + 'mult'] = mult
        keep[pfx + 'cat'] = cat
        keep[pfx + 'mlp3'] = m3
    return m3


def node_embedding(x, sd, keep=None):
    """models/blocks_emb.py:29-43: K chained blocks then max over the last index."""
    h = x
    for blk in range(1, num_blocks_of(sd) + 1):
        h = fgnn_block(h, sd, blk, keep)
    e = torch.max(h, -1)[0]
    if keep is not None:
        keep['ne/suffix'] = e
    return e


def siamese_scores(x1, x2, sd):
    """models/trainers.py:60-68."""
    e1 = node_embedding(x1, sd)
    e2 = node_embedding(x2, sd)
    return torch.matmul(torch.transpose(e1, 1, 2), e2)


def triplet_loss_mean(raw_scores):
    """toolbox/losses.py:20-34 with loss_reduction='mean' (a list of (n,n) works too)."""
    loss = 0
    total = 0
    for out in raw_scores:
        n_vertices = out.shape[0]
        target = torch.arange(n_vertices)
        loss += F.cross_entropy(out, target, reduction='sum')
        total += n_vertices
    return loss / total


def triplet_loss_mean_of_mean(raw_scores):
    """toolbox/losses.py:14-15,20-34 with loss_reduction='mean_of_mean': average over graphs of the per-graph mean CE."""
    loss = 0
    total = 0
    for out in raw_scores:
        n_vertices = out.shape[0]
        target = torch.arange(n_vertices)
        loss += F.cross_entropy(out, target, reduction='sum') / n_vertices
        total += 1
    return loss / total


# ----------------------------------------------------------------------------
# ragged batches: the reference's own tests define masked correctness as
# "list of per-graph dense runs" (maskedtensors/test_maskedtensor.py:22-27,141-150)
# ----------------------------------------------------------------------------
def node_embedding_ragged(x_list, sd):
    """x_list: list of (Cin, n_i, n_i).  Returns list of (C, n_i)."""
    return [node_embedding(x.unsqueeze(0), sd).squeeze(0) for x in x_list]


def siamese_scores_ragged(x1_list, x2_list, sd):
    e1 = node_embedding_ragged(x1_list, sd)
    e2 = node_embedding_ragged(x2_list, sd)
    return [torch.matmul(a.t(), b) for a, b in zip(e1, e2)]


def pad_graph_list(tensor_list):
    """maskedtensors/maskedtensor.py:8-48 restricted to dims=(1,2): zero-pad + 0/1 masks.

    Returns (data (B,C,Nmax,Nmax), n (B,) int64)."""
    nmax = max(t.size(-1) for t in tensor_list)
    c = tensor_list[0].size(0)
    data = torch.zeros((len(tensor_list), c, nmax, nmax), dtype=tensor_list[0].dtype)
    for i, t in enumerate(tensor_list):
        n = t.size(-1)
        data[i, :, :n, :n] = t
    return data, torch.tensor([t.size(-1) for t in tensor_list], dtype=torch.int64)


# ----------------------------------------------------------------------------
# whole training-step model work: forward both branches + loss + backward
# ----------------------------------------------------------------------------
def step_fwd_bwd(x1, x2, sd):
    """One step's model work.  Returns (scores, loss, grads{name: tensor})."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in _strip(sd).items()}
    scores = siamese_scores(x1, x2, params)
    loss = triplet_loss_mean(scores)
    names = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in names])
    return scores.detach(), loss.detach(), dict(zip(names, gs))


def step_fwd_bwd_ragged(x1_list, x2_list, sd):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in _strip(sd).items()}
    scores = siamese_scores_ragged(x1_list, x2_list, params)
    loss = triplet_loss_mean(scores)
    names = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in names])
    return [s.detach() for s in scores], loss.detach(), dict(zip(names, gs))


# ----------------------------------------------------------------------------
# parameter construction in the reference's init order (models/layers.py:118-123,
# 134-142, 63-66): xavier-uniform conv weights, zero biases, gn ones/zeros.
# Consumes the torch global RNG exactly like building the reference model does.
# ----------------------------------------------------------------------------
def init_state_dict(original_features_num=2, num_blocks=4, in_features=32,
                    out_features=32, depth_of_mlp=3, dtype=torch.float32):
    import torch.nn as nn
    sd = {}
    last = original_features_num
    for blk in range(1, num_blocks + 1):
        outf = in_features if blk < num_blocks else out_features
        for j, cin in ((1, last), (2, last), (3, last + outf)):
            pfx = 'ne_bm_block%d_mlp%d.' % (blk, j)
            c = cin
            for i in range(depth_of_mlp):
                conv = nn.Conv2d(c, outf, kernel_size=1, padding=0, bias=True)
                nn.init.xavier_uniform_(conv.weight)
                nn.init.zeros_(conv.bias)
                sd[pfx + 'convs.%d.weight' % i] = conv.weight.detach().to(dtype)
                sd[pfx + 'convs.%d.bias' % i] = conv.bias.detach().to(dtype)
                c = outf
            sd[pfx + 'gn.weight'] = torch.ones((1, outf, 1, 1), dtype=dtype)
            sd[pfx + 'gn.bias'] = torch.zeros((1, outf, 1, 1), dtype=dtype)
        last = outf
    return sd


def algorithmic_flops_per_pair(n, num_blocks=4, c=32, c0=2):
    """fwd+bwd = 3 x forward flops."""
    fg = 0
    cin = c0
    for _ in range(num_blocks):
        fg += 2 * n * n * (3 * cin * c + 7 * c * c) + 2 * n ** 3 * c
        cin = c
    return 3 * (2 * fg + 2 * n * n * c)


def max_rel_err(a, b):
    """max-norm relative error used throughout the parity tests."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    d = (a - b).abs().max().item() if a.numel() else 0.0
    s = b.abs().max().item() if b.numel() else 0.0
    return d / max(s, 1e-30) if s > 0 else d


if __name__ == '__main__':
    torch.manual_seed(0)
    sd = init_state_dict(num_blocks=1)
    x = torch.randn(2, 2, 8, 8)
    print(node_embedding(x, sd).shape, math.isfinite(float(node_embedding(x, sd).sum())))
