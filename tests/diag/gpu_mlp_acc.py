"""accuracy of the MlpBlock_Real HIP backward vs torch fp32 autograd, both against fp64 autograd"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graph_neural_net_amd.layers import MlpBlock_Real
from oracle import fgnn_oracle as O
DEV = 'cuda:0'
torch.manual_seed(0)
G, N, cin = 4, 50, int(sys.argv[2]) if len(sys.argv) > 2 else 32
mlp = MlpBlock_Real(cin, 32, 3).to(DEV)
with torch.no_grad():
    for c in mlp.convs: c.bias.add_(0.1 * torch.randn_like(c.bias))
    mlp.gn.weight.mul_(1 + 0.2 * torch.randn_like(mlp.gn.weight)); mlp.gn.bias.add_(0.05 * torch.randn_like(mlp.gn.bias))
x = torch.randn(G, cin, N, N)
mode = sys.argv[1] if len(sys.argv) > 1 else 'dense'
dy = torch.randn(G, 32, N, N)
if mode == 'sparse':      # like the pooling backward: one non-zero per row
    dy = torch.zeros(G, 32, N, N).scatter_(-1, torch.randint(0, N, (G, 32, N, 1)), torch.randn(G, 32, N, 1) * 1e-3)
xg = x.to(DEV).requires_grad_(True)
y = mlp(xg); y.backward(dy.to(DEV)); torch.cuda.synchronize()
ours = {k: p.grad.cpu() for k, p in mlp.named_parameters()}; ours['x'] = xg.grad.cpu()
def ref(dtype):
    ws = [c.weight.detach().cpu().to(dtype).requires_grad_(True) for c in mlp.convs]
    bs = [c.bias.detach().cpu().to(dtype).requires_grad_(True) for c in mlp.convs]
    gw = mlp.gn.weight.detach().cpu().to(dtype).requires_grad_(True); gb = mlp.gn.bias.detach().cpu().to(dtype).requires_grad_(True)
    xx = x.to(dtype).requires_grad_(True)
    out = O.mlp_block_real(xx, ws, bs, gw, gb); out.backward(dy.to(dtype))
    r = {'x': xx.grad, 'gn.weight': gw.grad, 'gn.bias': gb.grad}
    for i in range(3): r['convs.%d.weight' % i] = ws[i].grad; r['convs.%d.bias' % i] = bs[i].grad
    return r, out.detach()
r64, y64 = ref(torch.float64); r32, y32 = ref(torch.float32)
l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
print('forward: ours %.2e torch32 %.2e' % (l2(y.detach().cpu(), y64), l2(y32, y64)))
for k in sorted(r64):
    if k == 'convs.2.bias': continue
    print('%-16s ours %.2e  torch-fp32 %.2e' % (k, l2(ours[k], r64[k]), l2(r32[k], r64[k])))
