"""Single engine vs FgnnEngineDual (two half-batch chains on two streams): agreement and captured step time.
    python tools/gpu_dual_probe.py [B] [N] [reps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from graph_neural_net_amd.engine_dual import FgnnEngineDual

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)


def capture(work):
    work(); work()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    return g


def timeit(g, reps):
    for _ in range(64):
        g.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / reps * 1e3)
    return sorted(best)


g1 = torch.zeros_like(params)
e1 = FgnnEngine(lay, 2 * B, N, dev)
s1, l1 = e1.step(params, g1, x)
torch.cuda.synchronize()
g2 = torch.zeros_like(params)
e2 = FgnnEngineDual(lay, 2 * B, N, dev)
s2, l2 = e2.step(params, g2, x)
torch.cuda.synchronize()
print('scores max |diff| %.3e (max |s| %.3f)   loss %.7f vs %.7f   grads rel L2 diff %.3e'
      % ((s1 - s2).abs().max().item(), s1.abs().max().item(), l1.item(), l2.item(), ((g1 - g2).norm() / g1.norm()).item()))
s3, l3 = e2.step(params, g2.clone().zero_(), x)
gr = torch.zeros_like(params); e2.step(params, gr, x); torch.cuda.synchronize()
print('dual run-to-run bit-identical:', bool(torch.equal(gr, g2)))

gs = capture(lambda: e1.step(params, g1, x))
print('single engine, captured: ms/step (5 x %d replays) %s' % (reps, ['%.4f' % v for v in timeit(gs, reps)]))
e2.stage_inputs(x)
gd = capture(lambda: e2.step(params, g2, None))
print('dual engine,   captured: ms/step (5 x %d replays) %s' % (reps, ['%.4f' % v for v in timeit(gd, reps)]))
# half-CU workgroups without the second stream (what the co-scheduling is worth)
e3 = FgnnEngine(lay, 2 * B, N, dev, cu_share=2)
g3 = torch.zeros_like(params)
gh = capture(lambda: e3.step(params, g3, x))
print('single engine with half-CU workgroups, captured: %s' % (['%.4f' % v for v in timeit(gh, reps)]))
