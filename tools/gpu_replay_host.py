"""Is the replayed step host-bound?  For cfg2 (bit-packed input, structured block 1): host time of one graph.replay() call (no
synchronisation between the calls), device time per step, and the same with `r` steps recorded into ONE graph (the difference
per step = the cost of the graph boundary: launch of the next graph + the gap before its first kernel).
    gpurun -- python tools/gpu_replay_host.py [cfg2|cfg5|cfg4]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
nvalid = None
if cfg == 'cfg5':
    xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120, 'ErdosRenyi', 0.2, 0.1)
    sizes = [int(t.shape[-1]) for t in xs]; N = max(sizes); B = 8
    pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
    x1, x2 = pad(xs), pad(ys)
    nvalid = torch.tensor(sizes * 2, dtype=torch.int32, device=dev); tn = float(sum(sizes))
    eng = FgnnEngine(lay, 2 * B, N, dev, ragged=True, block1='structured')
elif cfg == 'cfg4':
    from graph_neural_net_amd.engine16 import FgnnEngineBF16
    B, N = 8, 200
    x1, x2 = synthetic.make_batch(4000, B, N, 'ErdosRenyi', 0.5, 0.1); tn = float(B * N)
    eng = FgnnEngineBF16(lay, 2 * B, N, dev, block1='structured')
else:
    B, N = 32, 50
    x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1); tn = float(B * N)
    eng = FgnnEngine(lay, 2 * B, N, dev, block1='structured')
bits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(dev)

def work():
    eng.step(params, grads, None, nvalid=nvalid, total_nodes=tn, bits=bits)

work(); torch.cuda.synchronize()

def capture(r):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(r):
            work()
    return g

for r in (1, 2, 4):
    g = capture(r)
    for _ in range(64 // r):
        g.replay()
    torch.cuda.synchronize()
    n = 200 // r
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        g.replay()
        host.append(time.perf_counter() - h0)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    host.sort()
    print('%s  %d step(s) per graph: device %.1f us per step; host replay() call median %.1f us, p90 %.1f us; all %d calls issued '
          'after %.1f ms of %.1f ms' % (cfg, r, t_all / n / r * 1e6, host[len(host) // 2] * 1e6, host[int(len(host) * 0.9)] * 1e6,
                                        n, t_issue * 1e3, t_all * 1e3))
    # one replay at a time (synchronise after each): the step without any overlap of launch and execution
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay(); torch.cuda.synchronize()
    print('      synchronised after every replay: %.1f us per step' % ((time.perf_counter() - t0) / 50 / r * 1e6))
