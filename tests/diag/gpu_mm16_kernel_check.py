"""Kernel-level check of fgnn_chan_matmul_fwd16 against torch on the rounded operands: python tests/diag/gpu_mm16_kernel_check.py N"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graph_neural_net_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 72
DEV = 'cuda:0'
G, Cc = 2, 32
ldr = (N + 7) // 8 * 8
ldp = (N * ldr + 63) // 64 * 64
gen = torch.Generator().manual_seed(N)
def slab(scale, shift):
    t = torch.zeros(G, Cc, ldp)
    v = torch.randn(G, Cc, N, N, generator=gen) * scale + shift
    t[:, :, :N * ldr].view(G, Cc, N, ldr)[..., :N] = v
    return t.to(torch.bfloat16).to(DEV).contiguous()
za, zb = slab(0.3, 1.5), slab(0.2, -0.7)
nrm_a = torch.zeros(G * Cc, 4); nrm_b = torch.zeros(G * Cc, 4)
nrm_a[:, 0], nrm_a[:, 1] = 1.5, 0.8
nrm_b[:, 0], nrm_b[:, 1] = -0.7, 1.3
nrm_a, nrm_b = nrm_a.to(DEV), nrm_b.to(DEV)
sa = _lib.make_slab16(za, Cc * ldp, ldp, Cc, nrm=nrm_a)
sb = _lib.make_slab16(zb, Cc * ldp, ldp, Cc, nrm=nrm_b)
out = torch.zeros(G * Cc * ldp, dtype=torch.bfloat16, device=DEV)
_lib.call('fgnn_chan_matmul_fwd16', C.byref(sa), C.byref(sb), None, G, N, ldr, _lib.ptr(out), Cc * ldp, ldp, _lib.stream_ptr())
dense = lambda t: t.view(G, Cc, ldp)[:, :, :N * ldr].reshape(G, Cc, N, ldr)[..., :N].float().cpu()
ya = ((dense(za) - 1.5) * 0.8).to(torch.bfloat16).float()
yb = ((dense(zb) + 0.7) * 1.3).to(torch.bfloat16).float()
ref = torch.matmul(ya.double(), yb.double()).float()
got = dense(out)
err = (got - ref).abs()
tol = 2.0 ** -7 * ref.abs() + 1e-6
bad = err > tol
print('N', N, 'bad', int(bad.sum()), 'of', bad.numel(), 'max err/tol', float((err / tol).max()))
idx = bad.nonzero()
if len(idx):
    print('rows', idx[:, 2].unique().tolist()[:40]); print('cols', idx[:, 3].unique().tolist()[:40])
    for i in idx[:8].tolist():
        print(i, got[tuple(i)].item(), ref[tuple(i)].item())
