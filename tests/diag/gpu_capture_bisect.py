"""Which piece of the generic-width module path survives HIP-graph capture?  Each case runs in its own process (a failing capture
can take the process down).  usage: python tests/diag/gpu_capture_bisect.py [case]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = ['conv', 'conv_bwd', 'gn', 'gn_bwd', 'mlp64', 'mlp64_bwd', 'matmul_bwd', 'colmax_bwd', 'loss_bwd', 'cat_bwd', 'model_fwd', 'model_full', 'fs_direct', 'fs_after_eager', 'fs_keep', 'fs_keep_cpu_inputs']


def run_case(name):
    import torch
    import torch.nn.functional as F
    from graph_neural_net_amd import layers
    from graph_neural_net_amd.siamese import Siamese_Node_Exp
    dev = 'cuda:0'
    torch.manual_seed(0)
    G, N = 4, 20
    x = torch.randn(G, 64, N, N, device=dev)
    if name.startswith('conv'):
        mod = torch.nn.Conv2d(64, 64, 1).to(dev)
        params = list(mod.parameters())
        fn = lambda: layers._ConvFn.apply(x, None, mod.weight, mod.bias, True)
    elif name.startswith('gn'):
        mod = layers.GraphNorm(64).to(dev)
        params = list(mod.parameters())
        fn = lambda: mod(x)
    elif name.startswith('mlp64'):
        mod = layers.MlpBlock_Real(64, 64, 3).to(dev)
        params = list(mod.parameters())
        fn = lambda: mod(x)
    elif name.startswith('matmul'):
        a = torch.randn(G, 8, N, N, device=dev, requires_grad=True)
        b = torch.randn(G, 8, N, N, device=dev, requires_grad=True)
        params = [a, b]
        fn = lambda: layers.Matmul()(a, b)
    elif name.startswith('colmax'):
        a = torch.randn(G, 8, N, N, device=dev, requires_grad=True)
        params = [a]
        fn = lambda: layers.ColumnMaxPooling()(a)
    elif name.startswith('loss'):
        from graph_neural_net_amd.losses import triplet_loss
        a = torch.randn(G, N, N, device=dev, requires_grad=True)
        params = [a]
        L = triplet_loss()
        fn = lambda: L(a)
    elif name.startswith('cat'):
        a = torch.randn(G, 8, N, N, device=dev, requires_grad=True)
        b = torch.randn(G, 8, N, N, device=dev, requires_grad=True)
        params = [a, b]
        fn = lambda: layers.Concat()(a, b)
    elif name.startswith('fs_'):
        ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=64,
                  out_features=64, depth_of_mlp=3)
        model = Siamese_Node_Exp(2, ne, metric='max').to(dev)
        x1, x2 = torch.randn(3, 2, N, N, device=dev), torch.randn(3, 2, N, N, device=dev)
        if name == 'fs_keep_cpu_inputs':
            g_ = torch.Generator().manual_seed(4)
            x1 = torch.randn(3, 2, N, N, generator=g_).to(dev)
            x2 = torch.randn(3, 2, N, N, generator=g_).to(dev)
        if name == 'fs_after_eager':
            model.loss(model(x1, x2)).backward()
        if name.startswith('fs_keep'):
            keep_scores = model(x1, x2)
            keep_loss = model.loss(keep_scores)
            keep_loss.backward()
            keep = {n: p.grad.clone() for n, p in model.named_parameters()}
            for p in model.parameters():
                p.grad = None
        l, s_ = model.fused_step(x1, x2)
        l2, _ = model.fused_step(x1, x2)
        torch.cuda.synchronize()
        print('CASE %s: captured and replayed, equal=%s' % (name, torch.equal(l, l2)))
        return
    else:
        ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=64,
                  out_features=64, depth_of_mlp=3)
        model = Siamese_Node_Exp(2, ne, metric='max').to(dev)
        x1, x2 = torch.randn(3, 2, N, N, device=dev), torch.randn(3, 2, N, N, device=dev)
        params = list(model.parameters())
        fn = (lambda: model(x1, x2)) if name == 'model_fwd' else (lambda: model.loss(model(x1, x2)))
    bwd = name.endswith('_bwd') or name == 'model_full'

    def run():
        for p in params:
            p.grad = None
        y = fn()
        if bwd:
            (y if y.dim() == 0 else y.sum()).backward()
        return y.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    g.replay()
    torch.cuda.synchronize()
    print('CASE %s: captured and replayed, equal=%s' % (name, torch.equal(out, ref)))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith('CASE')]
            print(line[0] if line else 'CASE %s: FAILED rc=%d %s' % (c, r.returncode, (r.stderr.strip().splitlines() or ['?'])[-1][:200]))
