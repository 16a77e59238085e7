#!/usr/bin/env python3
"""Benchmark of the 2-FGNN hot path: graph-pairs/sec, forward + loss + backward.

    python bench.py --gpus N --steps K --warmup W

One "step" = forward of both siamese branches + scoring + triplet loss + full backward to
all 96 parameter-gradient tensors on one batch of synthetic N=50 random-regular graph
pairs (BASELINE.json configs[1]: batch 32 per GPU, 4 blocks x 32 features, depth 3, fp32),
inputs resident in HBM.  With N > 1 ranks (torchrun, one process per GPU) every rank owns
its own 32-pair shard (weak scaling) and the step ends with ONE RCCL all-reduce of the flat
gradient buffer.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: RCCL (and any sharing of device tensors across processes) needs this
# before the first HIP call; the image exports it, a bare environment might not
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch                                                      # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from graph_neural_net_amd import _lib, dp, synthetic            # noqa: E402
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def kernel_model(tag, G, N, pix=None, cube=None):
    """Algorithmic (bytes, flops) of one launch of `tag` (interface reads + writes; fp32 slabs, bf16 for the *16 kernels).
    Ragged batches: pix = sum of n_g^2 and cube = sum of n_g^3 over the G graphs (the valid corners) replace G*N^2, G*N^3."""
    P = N * N
    if pix is not None:          # every formula below is linear in G*P and G*N^3
        P = pix / G
        N = (cube / G) ** (1.0 / 3.0)
    if tag.startswith('mlp_fwd16['):
        cin, nmlp = [int(v.split('=')[1]) for v in tag[10:-1].split(',')]
        return 2.0 * G * P * (cin + 32 * nmlp), 2.0 * G * P * nmlp * (cin * 32 + 2 * 1024)
    if tag.startswith('mlp_bwd16_pair['):
        cin, dx = [int(v.split('=')[1]) for v in tag[15:-1].split(',')]
        return 2.0 * G * P * (cin + 128 + 2 * dx), 2 * 4.0 * G * P * (cin * 32 + 2 * 1024)
    if tag.startswith('mlp_bwd16['):
        cin, dx = [int(v.split('=')[1]) for v in tag[10:-1].split(',')]
        return 2.0 * G * P * (cin + 64 + dx), 4.0 * G * P * (cin * 32 + 2 * 1024)
    if tag == 'fgnn_chan_matmul_fwd16':
        return 2.0 * G * 32 * P * 3, 2.0 * G * 32 * N ** 3
    if tag == 'fgnn_chan_matmul_bwd16':
        return 2.0 * G * 32 * P * 5, 4.0 * G * 32 * N ** 3
    if tag == 'fgnn_colmax_bwd16' or tag == 'fgnn_colmax_fwd16':
        return 2.0 * G * 32 * P, 1.0 * G * 32 * P
    if tag.startswith('mlp_fwd['):
        cin, nmlp = [int(v.split('=')[1]) for v in tag[8:-1].split(',')]
        return 4.0 * G * P * (cin + 32 * nmlp), 2.0 * G * P * nmlp * (cin * 32 + 2 * 1024)
    if tag.startswith('mlp_bwd_pair['):       # mlp1 + mlp2 of a block: x, 2 x (dy, z), the old d_in, one d_in store
        cin, dx = [int(v.split('=')[1]) for v in tag[13:-1].split(',')]
        return 4.0 * G * P * (cin + 128 + 2 * dx), 2 * 4.0 * G * P * (cin * 32 + 2 * 1024)
    if tag.startswith('mlp_bwd['):
        cin, dx = [int(v.split('=')[1]) for v in tag[8:-1].split(',')]
        return 4.0 * G * P * (cin + 64 + dx), 4.0 * G * P * (cin * 32 + 2 * 1024)
    if tag == 'fgnn_chan_matmul_fwd':
        return 4.0 * G * 32 * P * 3, 2.0 * G * 32 * N ** 3
    if tag == 'fgnn_chan_matmul_bwd':
        return 4.0 * G * 32 * P * 5, 4.0 * G * 32 * N ** 3
    if tag == 'fgnn_gn_bwd_stats':
        return 4.0 * G * 32 * P * 2, 4.0 * G * 32 * P
    if tag == 'fgnn_colmax_bwd' or tag == 'fgnn_colmax_fwd':
        return 4.0 * G * 32 * P, 1.0 * G * 32 * P
    return 0.0, 0.0


def algorithmic_per_pair(N, num_blocks=4, C=32, c0=2, elt=4):
    """(flops fwd+bwd, bytes fwd+bwd) per pair; elt = bytes per stored activation."""
    fg, by, cin = 0, 0, c0
    for _ in range(num_blocks):
        fg += 2 * N * N * (3 * cin * C + 7 * C * C) + 2 * N ** 3 * C
        by += 9 * cin + 17 * C
        cin = C
    return 3.0 * (2 * fg + 2 * N * N * C), 2.0 * elt * N * N * by


def executed_per_pair(N, num_blocks=4, C=32, c0=2, elt=4):
    """The same model for a step whose block 1 runs on its STRUCTURED input (csrc/block1_struct.hip): mlp1, mlp2 and the per-channel
    product of block 1 are not executed as dense work in either direction (class tables / closed form instead: integer and
    per-class work, counted as 0 flops) -- what remains of block 1 is the write of `mult`, mlp3, and in the backward direction
    mlp3's backward and one read of d(mult)."""
    fl, by = algorithmic_per_pair(N, num_blocks, C, c0, elt)
    fl -= 3.0 * 2 * (2 * N * N * 2 * (c0 * C + 2 * C * C) + 2 * N ** 3 * C)
    # bytes per pixel and graph, block 1: forward 2 (c0 + C) [mlp1, mlp2] + 3 C [product] -> C [mult written];
    # backward 2 (2 c0 + C) + 5 C -> C [d(mult) read]
    by -= 2.0 * elt * N * N * ((2 * (c0 + C) + 3 * C - C) + (2 * (2 * c0 + C) + 5 * C - C))
    return fl, by


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(layout, params, x1, x2, min_seconds=10.0, max_steps=400):
    """The oracle (pure PyTorch CPU, same ATen op sequence as the reference) on the host cores: thread sweep from 1 to
    all available cores on a 2-pair slice, then the best thread count timed on a bounded sample (~10 s of CPU work);
    the single-thread rate is reported next to it."""
    from oracle import fgnn_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sd = {k: v.clone() for k, v in layout.unflatten(params.cpu()).items()}
    sweep = {}
    best, cores = None, 1
    big = x1.shape[-1] > 64          # N = 200: seconds per pair -- one un-warmed 1-pair run per thread count
    sl = 1 if big else min(16, x1.shape[0])      # (a 2-pair slice under-rates the high thread counts: ATen parallelises over the batch)
    # the sweep goes up to 128 threads whatever the trend (measured on the 256-core box: 8-16 threads are the best, 128 threads are
    # ~20x slower -- oversubscribed ATen kernels on 50 x 50 planes); all `avail` cores are only tried while the rate has not
    # collapsed (256 threads took 200 s for two pairs)
    for c in sorted({min(avail, v) for v in ((1, 8, 32, 128) if big else (1, 4, 8, 16, 32, 64, 128, avail))}):
        if best is not None and c > 128 and sweep[max(sweep)] < 0.25 * (2.0 / best):
            break
        if big and best is not None and c > cores and sweep[max(sweep)] < 0.75 * (2.0 / best):
            break               # (N = 200: seconds per pair and thread count)
        torch.set_num_threads(c)
        if not big:
            O.step_fwd_bwd(x1[:sl], x2[:sl], sd)
        t0 = time.time()
        O.step_fwd_bwd(x1[:sl], x2[:sl], sd)
        dt = (time.time() - t0) * 2.0 / sl          # seconds per 2 pairs
        sweep[c] = 2.0 / dt
        if best is None or dt < best:
            best, cores = dt, c
    torch.set_num_threads(cores)
    # bounded sample: (a slice of) the same batch, repeated until ~10 s of CPU work have been timed
    pairs = x1.shape[0]
    est_full = best * pairs / 2.0
    if est_full > 10.0:
        pairs = max(1 if big else 2, int(pairs * 10.0 / est_full))
    xa, xb = x1[:pairs], x2[:pairs]
    t0 = time.time()
    n = 0
    while n < max_steps and (n == 0 or time.time() - t0 < min_seconds):
        O.step_fwd_bwd(xa, xb, sd)
        n += 1
    dt = (time.time() - t0) / n
    return {'value': pairs / dt, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'single_thread_value': sweep.get(1), 'cores_available': avail, 'cpu_model': _cpu_model(),
            'thread_sweep_pairs_per_s': {str(k): round(v, 2) for k, v in sweep.items()},
            'sample': '%d steps on %d of the %d pairs of the same batch (N=%d), %.2f s/step, torch %s CPU, fp32, '
                      '%d threads = best of the sweep %s (of %d available); single_thread_value and the sweep are rates on a '
                      '%d-pair slice' % (n, pairs, x1.shape[0], x1.shape[-1], dt, torch.__version__, cores,
                                        sorted(sweep), avail, sl)}


def cpu_baseline_ragged(layout, params, xs, ys, min_seconds=10.0, max_steps=100):
    """cfg5: the oracle on the ragged list, graph by graph (the reference's MaskedTensor path computes the padded batch;
    per-graph dense runs are its own tests' definition of the result and the cheaper CPU schedule)."""
    from oracle import fgnn_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sd = {k: v.clone() for k, v in layout.unflatten(params.cpu()).items()}
    sweep = {}
    for c in sorted({min(avail, v) for v in (1, 8, 16, 32)}):
        torch.set_num_threads(c)
        O.step_fwd_bwd_ragged(xs[:2], ys[:2], sd)
        t0 = time.time()
        O.step_fwd_bwd_ragged(xs[:2], ys[:2], sd)
        sweep[c] = 2.0 / (time.time() - t0)
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    t0 = time.time()
    n = 0
    while n < max_steps and (n == 0 or time.time() - t0 < min_seconds):
        O.step_fwd_bwd_ragged(xs, ys, sd)
        n += 1
    dt = (time.time() - t0) / n
    return {'value': len(xs) / dt, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'single_thread_value': sweep.get(1), 'cores_available': avail, 'cpu_model': _cpu_model(),
            'thread_sweep_pairs_per_s': {str(k): round(v, 2) for k, v in sweep.items()},
            'sample': '%d steps on the %d ragged pairs of the same batch, graph by graph, %.2f s/step, torch %s CPU, fp32, %d '
                      'threads = best of the sweep %s on the first 2 pairs' % (n, len(xs), dt, torch.__version__, cores, sorted(sweep))}


_COLLECTIVES = ('all_reduce', 'all_gather', 'all_gather_into_tensor', 'all_gather_object', 'broadcast', 'broadcast_object_list', 'reduce',
                'reduce_scatter', 'reduce_scatter_tensor', 'all_to_all', 'all_to_all_single', 'gather', 'scatter', 'send', 'recv',
                'isend', 'irecv', 'barrier')


def _count_collectives():
    """(--verify-dir) wrap every data-moving entry point of torch.distributed with a counter -> the list the names are appended to."""
    import torch.distributed as dist
    calls = []
    for name in _COLLECTIVES:
        fn = getattr(dist, name, None)
        if fn is None or getattr(fn, '_fgnn_counted', False):
            continue

        def wrapped(*a, _fn=fn, _name=name, **k):
            calls.append(_name)
            return _fn(*a, **k)
        wrapped._fgnn_counted = True
        setattr(dist, name, wrapped)
    return calls


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process -- which has NOT touched the GPU (no HIP call, no
    library load) -- starts N fresh rank processes of this script with the torchrun environment, relays rank 0's JSON
    line and exits with the worst return code.  (Never re-exec a process that has initialised the GPU.)"""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   FGNN_BENCH_CHILD='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # rank 0's stdout is drained by a thread (a full pipe would block it); ALL children are polled: when one dies with a
    # non-zero code (bad device index, RCCL init failure) the others -- stuck in the rendezvous or a barrier until the
    # distributed timeout -- are terminated and the parent reports at once
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            sys.stderr.write('bench.py: rank %d exited with code %d; stopping the other ranks\n' % (failed, rcs[failed]))
            for r, p in enumerate(procs):           # exactly the processes started above, by handle
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b''.join(chunks).decode('utf-8', 'replace'))
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(rcs[failed] if rcs[failed] > 0 else 1)
    raise SystemExit(max(rcs, key=abs))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='cfg2', choices=('cfg2', 'cfg4', 'cfg5'),
                    help='BASELINE.json configs[1] (N=50 regular pairs, batch 32, fp32; the headline line), '
                         'configs[3] (N=200 dense ER pairs, batch 8, bf16) or the per-GPU shard of configs[4] '
                         '(variable-N pairs, n in [30, 120], 8 pairs)')
    ap.add_argument('--precision', default=None, choices=('fp32', 'bf16'),
                    help="kernel set (default: the config's -- fp32 for cfg2, bf16 for cfg4); '--precision bf16' with cfg2 is the "
                         "reference's 16-bit training mode on the headline workload, reported with dtype bf16")
    ap.add_argument('--batch', type=int, default=None, help='pairs per GPU (default: the config\'s)')
    ap.add_argument('--n', type=int, default=None, help='vertices per graph (default: the config\'s)')
    ap.add_argument('--blocks', type=int, default=4)
    ap.add_argument('--path', default='engine', choices=('engine', 'module', 'fused_step'),
                    help="'engine': FgnnEngine.step (the fused launch sequence, the headline); 'module': the same batch through "
                         "the drop-in module surface -- Siamese_Node_Exp.forward, model.loss, loss.backward(), eager launches; "
                         "'fused_step': Siamese_Node_Exp.fused_step on the loader's batch (tensors or MaskedTensors): the module "
                         "surface with the step as ONE replayed graph")
    ap.add_argument('--module-input-form', default='dense', choices=('dense', 'tensor_representation'),
                    help="--path fused_step / module: the module's input_form ('tensor_representation': the dense loader batch is "
                         'bit-packed + verified on the device and block 1 runs on its structured form)')
    ap.add_argument('--no-graph', action='store_true', help='do not capture the step in a HIP graph')
    ap.add_argument('--chains', type=int, default=None, choices=(1, 2),
                    help='2: the batch runs as two half-batch chains on two streams / disjoint halves of the CUs (FgnnEngineDual: '
                         'about 2 %% faster at the default size, measured), 1 (default): one engine')
    ap.add_argument('--mfma', default=None, choices=('f32', 'x3'),
                    help="contraction of the fp32 MLP kernels: 'f32' (default) = v_mfma_f32_32x32x2_f32, 'x3' = bf16 matrix cores "
                         'through the exact three-way operand split (csrc/fgnn_x3.h)')
    ap.add_argument('--input', default=None, choices=('dense', 'bits'),
                    help="how the batch is handed to the engine: 'dense' = the (2B, 2, N, N) fp32 tensor representation, 'bits' = the "
                         'bit-packed adjacency (block 1 expands it itself: SURVEY 8 row f3).  Default: bits with --block1 structured, else dense')
    ap.add_argument('--block1', default=None, choices=('generic', 'structured'),
                    help="block 1 on bit-packed inputs: 'structured' = csrc/block1_struct.hip (class tables + closed-form per-channel "
                         "product; constant-size and ragged batches, N <= 256, fp32 and bf16 engines), 'generic' = the kernels every block uses")
    ap.add_argument('--settle', type=int, default=64, help='untimed replays before the warm-up steps (clock / TLB settling)')
    ap.add_argument('--windows', type=int, default=5,
                    help='the K-step timed window is repeated this many times; ms_per_step / value are the MEDIAN window, '
                         'ms_per_step_min / _max the spread')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='do not append the cfg4 / cfg5 measurements (extra_configs) to the headline line')
    ap.add_argument('--profile-steps', type=int, default=5, help='instrumented steps for the roofline leg')
    ap.add_argument('--trace-steps', type=int, default=0,
                    help='(diagnostic) after the timed windows: synchronize, then this many steps with an event between every two; the '
                         'per-step GPU times go to step_trace_ms (how a window starts after an idle device)')
    ap.add_argument('--verify-dir', default=None,
                    help='(tests) every rank counts the torch.distributed calls of each timed step and, after the timed windows, runs ONE '
                         'eager step + the all-reduce and writes DIR/rank<r>.pt: {collectives_per_step, comm = the all-reduced flat '
                         'gradient, loss_sum, nodes, sizes} -- compared with a single-process run on the concatenated batch')
    ap.add_argument('--backend', default=None, help="torch.distributed backend (default: nccl = RCCL); 'gloo' lets "
                                                    'several ranks share one GPU for functional tests')
    return ap.parse_args()


def main():
    if os.environ.get('FGNN_BENCH_CHILD') and os.environ.get('FGNN_BENCH_FAIL_RANK') == os.environ.get('RANK'):
        raise SystemExit(3)         # (tests: a rank that dies at start-up; the launcher must stop the job, see self_launch)
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args.gpus)                    # does not return
    rank, local_rank, world = dp.init_process_group(args.backend)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU; there is no CPU fallback for the product path')
    local_rank = local_rank % torch.cuda.device_count()     # (only differs in the shared-GPU functional test)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    out = run_config(args, args.config, rank, world, dev, cpu_leg=(world == 1 and not args.no_cpu_baseline))
    if rank == 0 and world == 1 and out['config'].get('block1', 'generic') != 'generic' and not args.no_extra_configs:
        # the same step on the DENSE tensor representation through the generic block-1 kernels, reported beside the headline
        sub = run_config(args, args.config, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3), block1='generic', input_form='dense')
        out['dense_input'] = {'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'],
                              'ms_per_step_min': sub['ms_per_step_min'], 'ms_per_step_max': sub['ms_per_step_max'],
                              'input': sub['config']['input'], 'block1': sub['config']['block1'],
                              'roofline': {k: sub['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')
                                           if sub['roofline'] and k in sub['roofline']}}
    if rank == 0 and world == 1 and args.config == 'cfg2' and args.path == 'engine' and args.mfma is None and args.precision is None \
            and not args.no_extra_configs:
        # the same step with mlp1 / mlp2 on the bf16 matrix cores through the exact three-way operand split (--mfma x3; opt-in: it
        # draws more power, the part runs at its package power limit and clocks ~4 % lower under it, so its gain -- 0 ... 45 us per
        # step -- depends on the box: DESIGN.md section 7, round 5); measured here so that the record holds both on the same box
        sub = run_config(args, args.config, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3), mfma='x3', profile=False)
        out['mfma_x3'] = {'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'], 'ms_per_step_min': sub['ms_per_step_min'],
                          'ms_per_step_max': sub['ms_per_step_max'], 'mlp_contraction': sub['config']['mlp_contraction']}
    if rank == 0 and world == 1 and args.config == 'cfg2' and args.path == 'engine' and args.mfma is None and args.precision is None \
            and args.batch is None and args.n is None and not args.no_extra_configs:
        # how the reference is actually run (side entries; the headline stays cfg2, fp32, 32 pairs per GPU):
        #   precision16 -- the cfg2 workload on the 16-bit engine (commander_explore.py:120-122 trains at precision=16; bf16 here)
        #   batch256    -- cfg2 at the reference's default_config.yaml batch (256 pairs) on one GPU
        sub = run_config(args, args.config, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3), profile=False, precision_override='bf16')
        out['precision16'] = {'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'], 'ms_per_step_min': sub['ms_per_step_min'],
                              'ms_per_step_max': sub['ms_per_step_max'], 'dtype': sub['dtype'], 'batch_per_gpu': sub['config']['batch_per_gpu'],
                              'workload': sub['config']['workload'], 'block1': sub['config'].get('block1'), 'input': sub['config'].get('input')}
        sub = run_config(args, args.config, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3), profile=False, batch_override=256)
        out['batch256'] = {'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'], 'ms_per_step_min': sub['ms_per_step_min'],
                           'ms_per_step_max': sub['ms_per_step_max'], 'dtype': sub['dtype'], 'batch_per_gpu': sub['config']['batch_per_gpu'],
                           'workload': sub['config']['workload'], 'block1': sub['config'].get('block1'), 'input': sub['config'].get('input')}
        #   wide64      -- the cfg2 workload on the 64-feature model (the reference takes any widths: models/layers.py:113-123)
        out['wide64'] = wide64_leg(args, dev)
    if rank == 0 and world == 1 and args.path == 'engine' and not args.no_extra_configs and out['config'].get('block1', 'generic') != 'generic':
        # ... and through the module surface a user of the reference calls (dense loader batch in, fused_step)
        out['module_surface'] = module_surface_leg(args, args.config, rank, world, dev, out['ms_per_step'])
    if rank == 0 and world == 1 and args.config == 'cfg2' and args.precision is None and args.path == 'engine' \
            and not args.no_extra_configs and args.batch is None and args.n is None:
        # BASELINE configs 4 and 5 (per-GPU shard) under the same K / W protocol, outside the headline's timed region
        extra = {}
        for cfg in ('cfg4', 'cfg5'):
            t0 = time.time()
            sub = run_config(args, cfg, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3))
            extra[cfg] = {'metric': sub['metric'], 'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'],
                          'ms_per_step_min': sub['ms_per_step_min'], 'ms_per_step_max': sub['ms_per_step_max'],
                          'dtype': sub['dtype'], 'workload': sub['config']['workload'],
                          'roofline': {k: sub['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')
                                       if sub['roofline'] and k in sub['roofline']},
                          'step_model': sub['step_model']}
            if cfg == 'cfg5':
                extra[cfg]['module_surface'] = module_surface_leg(args, cfg, rank, world, dev, sub['ms_per_step'])
            extra[cfg]['wall_s'] = round(time.time() - t0, 2)
        out['extra_configs'] = extra
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dp.barrier()            # rank 0 is still in its roofline leg while the others are done
        torch.distributed.destroy_process_group()


def time_in_graph(tag, engines, work, min_launches=24, replays=20):
    """Average duration of the launches tagged `tag` when issued back to back inside a replayed HIP graph.  Every engine runs one full
    step first (so its workspaces hold a step's real tensors), recording its launches; the graph then cycles through the recorded
    launches of `tag` -- every block's, every engine's: distinct operand sets -- until at least `min_launches` are in it."""
    sets = []
    for e in engines:
        work(e)                                   # allocations, kernel attributes
        torch.cuda.synchronize()
        _lib.PROFILE = []
        work(e)
        torch.cuda.synchronize()
        rec, _lib.PROFILE = _lib.PROFILE, None
        sets += [(name, a) for t, _, _, name, a in rec if t == tag]
    if not sets:
        raise RuntimeError('no launch tagged %s' % tag)
    rounds = -(-min_launches // len(sets))

    def burst():
        for _ in range(rounds):
            for name, a in sets:
                _lib.relaunch(name, a)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        burst()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):      # (a process group's watchdog thread may query events meanwhile)
        burst()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    n = rounds * len(sets)
    return {'ms': e0.elapsed_time(e1) / (replays * n), 'launches': n, 'replays': replays, 'sets': len(sets), 'engines': len(engines)}


def time_in_step(tag, eng, work, replays=40, rounds=5):
    """What the launches tagged `tag` cost INSIDE the step: the step's own launch sequence (every C call of one engine step, recorded and
    re-issued in order) is captured twice -- as it is, and with every launch of `tag` issued twice in a row -- and both graphs are
    replayed alternately; (t_doubled - t_plain) / launches is the kernel's duration in its real surroundings (the cache state the
    preceding kernel leaves, the clock the whole step runs at), which a burst of back-to-back launches of the step's most power-hungry
    kernel does not reproduce (time_in_graph: 60.4 - 62.5 us on boxes where the kernel trace of the replayed step shows 59.2).
    The second launch of a pair re-reads what the first one read and accumulates into the same d_in: same traffic, same arithmetic."""
    work(eng)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    work(eng)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    launches = [(name, a, t == tag) for t, _, _, name, a in rec]
    ndom = sum(1 for _, _, d in launches if d)
    if not ndom:
        raise RuntimeError('no launch tagged %s' % tag)

    def run(double):
        for name, a, dom in launches:
            _lib.relaunch(name, a)
            if double and dom:
                _lib.relaunch(name, a)
    graphs = []
    for double in (False, True):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(double)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            run(double)
        graphs.append(g)
    for g in graphs:
        for _ in range(5):
            g.replay()
    torch.cuda.synchronize()
    diffs, plain = [], []
    for _ in range(rounds):
        ms = []
        for g in graphs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / replays)
        plain.append(ms[0])
        diffs.append((ms[1] - ms[0]) / ndom)
    diffs.sort()
    return {'ms': diffs[len(diffs) // 2], 'ms_min': diffs[0], 'ms_max': diffs[-1], 'launches_per_step': ndom, 'replays': replays, 'rounds': rounds,
            'step_ms_relaunched': sorted(plain)[len(plain) // 2]}


def wide64_leg(args, dev):
    """The cfg2 workload (N = 50 regular pairs, 32 pairs) on the 64-FEATURE model (original_features_num 2, in_features = out_features = 64,
    depth 3, 4 blocks; models/layers.py:113-123 takes any widths) through the module surface: Siamese_Node_Exp.fused_step = forward +
    loss + backward of the per-module path captured into ONE replayed HIP graph.  Its conv stacks run on the fused 64-wide kernels of
    csrc/mlp64.hip; GraphNorm, per-channel products and loss are per-module launches.  Same K / W protocol."""
    from graph_neural_net_amd.siamese import Siamese_Node_Exp
    B, N = 32, 50
    x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
    x1, x2 = x1.to(dev), x2.to(dev)
    torch.manual_seed(64)
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=args.blocks, in_features=64, out_features=64,
              depth_of_mlp=3)
    model = Siamese_Node_Exp(2, ne, metric='max').to(dev)

    def step():
        model.fused_step({'input': x1}, {'input': x2})
    for _ in range(3 + args.warmup):
        step()
    torch.cuda.synchronize()
    window_s = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        window_s.append(time.perf_counter() - t0)
    ms = sorted(window_s)[1] / args.steps * 1e3
    return {'value': B / (ms * 1e-3), 'unit': 'pairs/s', 'ms_per_step': ms, 'ms_per_step_min': min(window_s) / args.steps * 1e3,
            'ms_per_step_max': max(window_s) / args.steps * 1e3, 'dtype': 'f32', 'batch_per_gpu': B, 'features': 64,
            'workload': 'N=50 regular-graph pairs, batch=%d, %d blocks, 2 -> 64 -> 64 features, depth 3' % (B, args.blocks),
            'path': 'Siamese_Node_Exp.fused_step: the per-module path in one replayed HIP graph; conv stacks on csrc/mlp64.hip'}


def module_surface_leg(args, config, rank, world, dev, engine_ms):
    """The same workload through the drop-in module surface (models/trainers.py:60-76): `Siamese_Node_Exp.fused_step` on the DENSE
    loader batch (tensors, or MaskedTensors for cfg5), same K / W protocol, with the module's input_form opt-in
    ('tensor_representation': the batch is bit-packed + verified on the device inside the step and block 1 runs structured) and
    without it ('dense': generic block 1)."""
    leg = {}
    for form in ('tensor_representation', 'dense'):
        sub = run_config(args, config, rank, world, dev, cpu_leg=False, windows=min(args.windows, 3), path='fused_step',
                         module_input_form=form, profile=False)
        leg[form] = {'value': sub['value'], 'unit': sub['unit'], 'ms_per_step': sub['ms_per_step'],
                     'ms_per_step_min': sub['ms_per_step_min'], 'ms_per_step_max': sub['ms_per_step_max'],
                     'block1': sub['config']['block1'], 'input': sub['config']['input']}
    leg['ms_per_step'] = leg['tensor_representation']['ms_per_step']
    leg['engine_ms_per_step'] = engine_ms
    leg['vs_engine'] = leg['ms_per_step'] / engine_ms
    leg['what'] = ("Siamese_Node_Exp(input_form='tensor_representation').fused_step({'input': x1}, {'input': x2}) on the dense loader "
                   'batch: ONE pack launch for both sides (fgnn_pack_adjacency_pair, with the device verdict) + ONE replayed HIP graph per step')
    return leg


def run_config(args, config, rank, world, dev, cpu_leg, windows=None, block1=None, input_form=None, path=None,
               module_input_form=None, profile=True, mfma=None, precision_override=None, batch_override=None):
    """One measurement: build the workload of `config`, capture the step, settle, warm up, time `windows` windows of
    exactly K steps (barrier + synchronize on both sides, max over ranks), roofline leg.  Returns the JSON dict (rank 0).
    block1 / input_form: override --block1 / --input (the dense-input line reported beside the headline); path /
    module_input_form: override --path / --module-input-form (module_surface_leg)."""
    windows = args.windows if windows is None else windows
    mfma_arg = mfma
    precision = precision_override if precision_override is not None else (args.precision if config == args.config else None)
    dense_er = config == 'cfg4'                   # the workload
    ragged = config == 'cfg5'                     # variable-N pairs, n in [30, N], one batch padded to its largest graph
    bf16 = (precision == 'bf16') if precision else dense_er      # the kernel set
    same = config == args.config
    B = batch_override if batch_override is not None else (args.batch if (same and args.batch is not None) else (8 if (dense_er or ragged) else 32))
    N = args.n if (same and args.n is not None) else (200 if dense_er else (120 if ragged else 50))
    path = path if path is not None else (args.path if same else 'engine')
    mform = module_input_form if module_input_form is not None else args.module_input_form
    no_graph = args.no_graph
    layout = ParamLayout(2, args.blocks, 32, 32, 3)
    params = layout.init_flat(0, dev)
    grads = torch.zeros_like(params)
    nvalid, pix, cube, sizes = None, None, None, None
    xs = ys = None
    if ragged:     # cfg5: Erdos-Renyi pairs (edge density 0.2, ER edge noise 0.1), n ~ U{30..N}
        xs, ys = synthetic.make_ragged_batch(5000 + rank, B, 30, N, 'ErdosRenyi', 0.2, 0.1)
        sizes = [int(t.shape[-1]) for t in xs]
        N = max(sizes)
        pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
        x1, x2 = pad(xs), pad(ys)
        nvalid = torch.tensor(sizes * 2, dtype=torch.int32, device=dev)
        pix, cube = 2.0 * sum(n * n for n in sizes), 2.0 * sum(n ** 3 for n in sizes)
    elif dense_er:   # cfg4: dense Erdos-Renyi (edge density 0.5) pairs, ER edge noise 0.1
        x1, x2 = synthetic.make_batch(4000 + rank, B, N, 'ErdosRenyi', 0.5, 0.1)
    else:
        x1, x2 = synthetic.make_batch(2000 + rank, B, N, 'Regular', 0.2, 0.1)
    def make_engine():
        if bf16:
            from graph_neural_net_amd.engine16 import FgnnEngineBF16
            # cfg4 (16-bit engine) takes the same default as the fp32 lines: bit-packed adjacency + structured block 1
            b1 = block1 if block1 is not None else (args.block1 if same else None)
            if b1 is None:
                b1 = 'structured' if (not (same and args.input == 'dense') and path == 'engine') else 'generic'
            return FgnnEngineBF16(layout, 2 * B, N, dev, ragged=ragged, block1=b1)
        chains = args.chains if (same and args.chains is not None) else 1
        mfma = mfma_arg if mfma_arg is not None else (args.mfma if (same and args.mfma is not None) else None)
        if chains == 2:
            from graph_neural_net_amd.engine_dual import FgnnEngineDual
            return FgnnEngineDual(layout, 2 * B, N, dev, ragged=ragged, mfma=mfma)
        # The headline line (cfg2, fp32 engine) hands the batch over as bit-packed adjacency and runs block 1 on its structured
        # form unless told otherwise; `dense_input` in the JSON is the same step on the dense tensor through the generic kernels
        b1 = block1 if block1 is not None else (args.block1 if same else None)
        if b1 is None:      # cfg2 and cfg5 (fp32 engine, N <= 128): structured unless the dense input was asked for
            b1 = 'structured' if (config in ('cfg2', 'cfg5') and not (same and args.input == 'dense') and path == 'engine') else 'generic'
        return FgnnEngine(layout, 2 * B, N, dev, ragged=ragged, mfma=mfma, block1=b1)
    eng = make_engine()
    x = torch.cat([x1, x2]).contiguous().to(dev)
    struct1 = bool(getattr(eng, 'struct1', False))
    want = input_form if input_form is not None else (args.input if same else None)
    use_bits = (path == 'engine' and not hasattr(eng, 'stage_inputs') and ((want == 'bits' and (struct1 or not bf16)) if want is not None else struct1))
    xbits = None
    if use_bits:     # the same batch as 32-bit words of adjacency rows (synthetic.pack_adjacency), resident in HBM like x
        import numpy as np
        xbits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(dev)
    dual = hasattr(eng, 'stage_inputs')
    if dual:
        eng.stage_inputs(x, nvalid)        # loader work, like the cat above: the chains' input buffers are resident before the timed region
    # loss normaliser of the concatenated global batch (toolbox/losses.py:27-34): ranks of a ragged batch hold different node
    # counts, so the global count is summed over the ranks once at set-up (not a collective of the step)
    total_nodes = dp.global_node_count(sum(sizes), dev) if ragged else float(B * N * world)

    def model_work(eng=eng):
        if xbits is not None:
            eng.step(params, grads, None, nvalid=nvalid, total_nodes=total_nodes, bits=xbits)
        else:
            eng.step(params, grads, None if dual else x, nvalid=None if dual else nvalid, total_nodes=total_nodes)
    engine_work = model_work

    model = None
    if path in ('module', 'fused_step'):
        # the surface a user of the reference calls (models/trainers.py:60-76): same weights, same batch, eager launches
        from graph_neural_net_amd.siamese import Siamese_Node_Exp
        node_emb = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=args.blocks,
                        in_features=32, out_features=32, depth_of_mlp=3)
        if ragged:
            node_emb['constant_n_vertices'] = False
        model = Siamese_Node_Exp(2, node_emb, metric='max', precision='bf16' if bf16 else 'fp32', input_form=mform).to(dev)
        with torch.no_grad():
            for (name, off, shape), (_, p) in zip(layout.entries, model.node_embedder.named_parameters()):
                p.copy_(params[off:off + p.numel()].view(shape))
        xa, xb = x[:B], x[B:]
        if ragged:      # what a user of the reference builds: one MaskedTensor batch per side (maskedtensor.from_list)
            from graph_neural_net_amd.masked import from_list
            xa = from_list([t.to(dev) for t in xs], dims=(1, 2), base_name='N')
            xb = from_list([t.to(dev) for t in ys], dims=(1, 2), base_name='M')
        no_graph = True

        def model_work():                      # noqa: F811
            if path == 'fused_step':
                model.fused_step({'input': xa}, {'input': xb})
                return
            for p in model.parameters():
                p.grad = None
            loss = model.loss(model({'input': xa}, {'input': xb})) * ((sum(sizes) if ragged else B * N) / total_nodes)
            loss.backward()

    graph = None
    model_work()                                  # allocates the backward workspace, sets kernel attributes
    torch.cuda.synchronize()
    # did the module surface put this batch on the structured block 1?  (fused_step's engine of this shape says so)
    surface_struct = bool(model is not None and path == 'fused_step' and any(
        getattr(e, 'struct1', False) and getattr(e, '_step_state', None) is not None for e in model.node_embedder._engines.values()))
    comm = grads if path == 'engine' else model.node_embedder._flat_grad
    # the ONE collective of a step: recorded into the step's HIP graph when the backend can be captured (RCCL), so a step is one
    # replay with no host launch on its critical path; otherwise (gloo, --no-graph, module path) an eager call after the model work
    ar_in_graph = world > 1 and not no_graph and dp.collective_captures()
    if world > 1:
        dp.warm_up_collective(dev)                # communicator set-up: not inside a capture, not inside the timed region
    if not no_graph:
        def capture(with_collective):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model_work()
                if with_collective:
                    dp.allreduce_sum_(comm)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # with a collective in the capture the process group's watchdog thread is alive: its event queries must not be
            # counted against this thread's capture ('thread_local'; the default 'global' mode would invalidate the capture)
            with torch.cuda.graph(g, capture_error_mode='thread_local' if with_collective else 'global'):
                model_work()
                if with_collective:
                    dp.allreduce_sum_(comm)
            return g
        try:
            graph = capture(ar_in_graph)
        except Exception as exc:                  # first fall back to a graph of the model work + an eager all-reduce ...
            graph = None
            if rank == 0:
                print('bench.py: HIP graph capture%s failed (%s)' % (' with the all-reduce inside' if ar_in_graph else '', exc), file=sys.stderr)
            torch.cuda.synchronize()
            if ar_in_graph:
                ar_in_graph = False
                try:
                    graph = capture(False)
                except Exception as exc2:         # ... then to eager launches (still the HIP path)
                    graph = None
                    if rank == 0:
                        print('bench.py: HIP graph capture failed (%s); running eager' % (exc2,), file=sys.stderr)
                    torch.cuda.synchronize()

    calls, per_step = None, []
    if args.verify_dir and same:
        calls = _count_collectives()

    def step():
        n0 = len(calls) if calls is not None else 0
        if graph is not None:
            graph.replay()
        else:
            model_work()
        if world > 1 and not ar_in_graph:
            dp.allreduce_sum_(comm)
        if calls is not None:
            per_step.append(calls[n0:])

    # Settling (untimed, part of the set-up like the capture runs above): the first ~20 replays after the set-up phase run
    # ~4 % slower than steady state (clocks, TLBs; `--steps 20 --warmup 5` gave 0.945 ms against 0.905 ms for any longer
    # run), and the metric is defined on the steady state.  Reported as config.settle_steps.
    for _ in range(args.settle):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # `windows` timed windows of EXACTLY K steps each, every one bracketed by barrier + synchronize on both sides and
    # reduced with MAX over the ranks; the line reports the median window (and the spread).
    window_s = []
    for _ in range(max(1, windows)):
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        window_s.append(dp.max_over_ranks(time.perf_counter() - t0, dev))
    elapsed = sorted(window_s)[len(window_s) // 2]
    # the collective on its own (outside the timed windows: events inside them would cost the step two host calls each):
    # 20 back-to-back eager all-reduces of the same buffer on every rank, event-timed on rank 0's stream
    allreduce_ms = None
    if world > 1:
        dp.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dp.allreduce_sum_(comm)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / 20

    step_trace = None
    if args.trace_steps > 0 and same:
        torch.cuda.synchronize()
        time.sleep(0.002)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.trace_steps + 1)]
        evs[0].record()
        for i in range(args.trace_steps):
            step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        step_trace = [round(evs[i].elapsed_time(evs[i + 1]), 4) for i in range(args.trace_steps)]
    if calls is not None:       # (tests) one eager step + the collective, dumped by every rank
        torch.cuda.synchronize()
        comm.zero_()
        model_work()
        loss_local = float(eng._loss_target.item()) if path == 'engine' and not dual else None
        if world > 1:
            dp.allreduce_sum_(comm)
        torch.cuda.synchronize()
        os.makedirs(args.verify_dir, exist_ok=True)
        torch.save({'collectives_per_step': per_step[-args.steps * max(1, windows):], 'comm': comm.detach().cpu().clone(),
                    'loss_local': loss_local, 'total_nodes': total_nodes, 'sizes': sizes, 'world': world, 'rank': rank,
                    'allreduce_in_graph': bool(ar_in_graph), 'batch_per_gpu': B, 'n': N,
                    'input': 'bits' if xbits is not None else 'dense', 'struct1': struct1},
                   os.path.join(args.verify_dir, 'rank%d.pt' % rank))

    # ---- roofline leg: per-kernel durations from events on the launch stream (eager launches) ----
    roofline = None
    kernels = {}
    if rank == 0 and args.profile_steps > 0 and not dual and profile:
        model_work()                              # one eager step untimed: the replays before it ran from the graph's own launch path
        torch.cuda.synchronize()
        _lib.PROFILE = []
        for _ in range(args.profile_steps):
            model_work()
        torch.cuda.synchronize()
        rec, _lib.PROFILE = _lib.PROFILE, None
        samples = {}
        for tag, e0, e1, _, _ in rec:
            samples.setdefault(tag, []).append(e0.elapsed_time(e1))
        # Average launch duration per kernel, robust against a stall of the box: a sample beyond 3 x the kernel's median is not
        # that kernel's duration (observed once: ONE 39 ms sample of a 10 us pooling launch made it the "dominant kernel" of the
        # line) -- such samples are dropped and counted (`outliers_dropped`)
        dropped = 0
        for tag, d in samples.items():
            med = sorted(d)[len(d) // 2]
            kept = [v for v in d if v <= 3.0 * med] or d
            dropped += len(d) - len(kept)
            kernels[tag] = [len(d), sum(kept) / len(kept) * len(d)]
        tot = sum(v[1] for v in kernels.values())
        summary = {t: {'launches_per_step': v[0] / args.profile_steps, 'avg_ms': v[1] / v[0],
                       'share': v[1] / tot} for t, v in kernels.items()}
        dom = max(kernels, key=lambda t: kernels[t][1])
        by, fl = kernel_model(dom, 2 * B, N, pix, cube)
        eager_ms = summary[dom]['avg_ms']
        # What that kernel costs INSIDE the replayed step (eager launches with events around them run 5-10 % longer: launch gaps, another
        # clock state): >= 24 back-to-back launches of the dominant kernel in a replayed HIP graph, cycling through every launch of that
        # kernel in the step (blocks 4, 3, 2 ...: distinct operand sets of this engine's workspace, > 400 MB in rotation at cfg2) -- the
        # protocol of tools/gpu_mm_ablate.py.  (Rotating THREE engines' workspaces -- nothing warm anywhere -- measures 70.5 us where the
        # kernel trace of the replayed step shows 67.6: in the step dy1 / dy2 were written by the launch before.)  Falls back to the
        # eager figure if the capture fails.
        ingraph = None
        if path == 'engine':
            try:
                ingraph = time_in_graph(dom, [eng], engine_work)
            except Exception as exc:          # noqa: BLE001 -- the eager figure stands
                print('bench.py: in-graph timing of %s failed (%s); roofline from eager events' % (dom, exc), file=sys.stderr)
                torch.cuda.synchronize()
        instep = None
        if path == 'engine' and ingraph is not None:
            try:
                instep = time_in_step(dom, eng, engine_work)
            except Exception as exc:          # noqa: BLE001 -- the back-to-back figure stands
                print('bench.py: in-step timing of %s failed (%s); roofline from the back-to-back launches' % (dom, exc), file=sys.stderr)
                torch.cuda.synchronize()
        dur = (instep['ms'] if instep else (ingraph['ms'] if ingraph else eager_ms)) * 1e-3
        mfma_peak = MFMA_BF16_PEAK_TF if bf16 else MFMA_F32_PEAK_TF
        if fl / (mfma_peak * 1e12) >= by / (HBM_PEAK_GBS * 1e9):
            ach = fl / dur / 1e12
            roofline = {'bound': 'mfma', 'achieved': ach, 'peak': mfma_peak, 'unit': 'TFLOP/s',
                        'frac': ach / mfma_peak, 'traffic': None}
        else:
            ach = by / dur / 1e9
            roofline = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBS, 'traffic': None}
        try:   # HBM traffic of that kernel from the committed PMC passes (same workload), if present
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
                tr = json.load(f)
            if ragged:      # cfg5: the PMC passes ran on exactly this batch (seed 5000, 8 pairs, n in [30, 120], fp32, one rank)
                key = 'cfg5:' + dom
                ok = (B, args.n or 120) == (8, 120) and not bf16 and rank == 0
            else:
                key = dom if not bf16 else 'cfg4:' + dom
                ok = (B, N) == ((8, 200) if bf16 else (32, 50)) and bf16 == dense_er
            dom_key = key if (key in tr and ok) else None
            if dom_key is not None:
                roofline['traffic'] = tr[dom_key]['bytes']
                roofline['traffic_source'] = tr.get('_source')
        except (OSError, ValueError):
            pass
        roofline.update({'kernel': dom, 'avg_launch_ms': dur * 1e3, 'in_graph_launch_ms': (dur * 1e3 if (ingraph or instep) else None), 'eager_launch_ms': eager_ms,
                         'in_step_launch_ms': instep['ms'] if instep else None,
                         'in_step_spread_ms': [instep['ms_min'], instep['ms_max']] if instep else None,
                         'back_to_back_launch_ms': ingraph['ms'] if ingraph else None,
                         'timing': (('the replayed step with every launch of this kernel issued twice minus the replayed step, per launch (%d launches per '
                                     'step, %d x %d replays of each graph, median); back_to_back_launch_ms: ' % (instep['launches_per_step'], instep['rounds'],
                                                                                                               instep['replays'])) if instep else '') +
                                   (('%d back-to-back launches per replay of a HIP graph (x %d replays), cycling %d operand sets of %d engines'
                                     % (ingraph['launches'], ingraph['replays'], ingraph['sets'], ingraph['engines'])) if ingraph else
                                    'eager launches, events around each'),
                         'outliers_dropped': dropped,
                         'algorithmic_bytes_per_launch': by, 'algorithmic_flops_per_launch': fl,
                         'alt_hbm_gbs': by / dur / 1e9, 'alt_mfma_tflops': fl / dur / 1e12})
        kernels = summary
        # context for the fractions above: what a cache-cold device copy moves on THIS box (the best case of a streaming
        # kernel whose inputs another kernel wrote) -- 8 x (32 MiB -> 32 MiB), event-timed
        try:
            bufs = [(torch.empty(8 << 20, dtype=torch.float32, device=dev).normal_(), torch.empty(8 << 20, dtype=torch.float32, device=dev))
                    for _ in range(8)]
            for s_, d_ in bufs:
                d_.copy_(s_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(4):
                for s_, d_ in bufs:
                    d_.copy_(s_)
            e1.record()
            torch.cuda.synchronize()
            roofline['cold_copy_gbs'] = 4 * 8 * 2 * (32 << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del bufs
        except RuntimeError:
            pass

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        fl_pair, by_pair = algorithmic_per_pair(N, args.blocks, elt=2 if bf16 else 4)
        struct_ran = bool((struct1 and xbits is not None) or surface_struct)
        fl_exec, by_exec = executed_per_pair(N, args.blocks, elt=2 if bf16 else 4) if struct_ran else (fl_pair, by_pair)
        if ragged:      # mean over the pairs of this batch, each at its own size
            per = [algorithmic_per_pair(n, args.blocks) for n in sizes]
            fl_pair, by_pair = sum(p[0] for p in per) / B, sum(p[1] for p in per) / B
            per = [executed_per_pair(n, args.blocks) if struct_ran else algorithmic_per_pair(n, args.blocks) for n in sizes]
            fl_exec, by_exec = sum(p[0] for p in per) / B, sum(p[1] for p in per) / B
            workload = ('cfg5: variable-N Erdos-Renyi pairs (edge density 0.2, ER edge noise 0.1), n ~ U{30..%d} (this batch: %s), '
                        '%d pairs per GPU in ONE batch padded to its largest graph, %d FGNN blocks x 32 features, depth 3, '
                        'siamese fwd + triplet loss + bwd' % (args.n or 120, sizes, B, args.blocks))
        elif dense_er:
            workload = ('cfg4: N=%d dense Erdos-Renyi pairs (edge density 0.5, ER edge noise 0.1), %d pairs per GPU, '
                        '%d FGNN blocks x 32 features, depth 3, siamese fwd + triplet loss + bwd' % (N, B, args.blocks))
        else:
            workload = ('cfg2: N=%d random-regular pairs (d=%d, ER edge noise 0.1), %d pairs per GPU, '
                        '%d FGNN blocks x 32 features, depth 3, siamese fwd + triplet loss + bwd'
                        % (N, synthetic.regular_degree(N, 0.2), B, args.blocks))
        if bf16:
            workload += ', bf16 storage + bf16 MFMA, fp32 accumulation / statistics / gradients'
        out = {
            'metric': ('graph-pairs/sec FGNN fwd+bwd, variable-N pairs (n in [30, %d])%s' % (args.n or 120, ', bf16' if bf16 else '')) if ragged else
                      'graph-pairs/sec FGNN fwd+bwd, N=%d %s pairs%s' % (N, 'dense ER' if dense_er else 'regular', ', bf16' if bf16 else ''),
            'value': value, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'ms_per_step_min': min(window_s) / args.steps * 1e3, 'ms_per_step_max': max(window_s) / args.steps * 1e3,
            'timed_windows': len(window_s), 'window_ms_per_step': [round(w / args.steps * 1e3, 5) for w in window_s], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': workload,
                       'batch_per_gpu': B, 'global_batch': B * world, 'n_vertices': N, 'num_blocks': args.blocks,
                       'parallelism': 'dp%d' % world, 'hip_graph': graph is not None, 'allreduce_in_graph': bool(ar_in_graph), 'path': path, 'settle_steps': args.settle,
                       'chains': 2 if dual else 1,
                       'input': ('bit-packed adjacency (N x ceil(N/32) words per graph)' if xbits is not None else
                                 ('dense (2, N, N) fp32 loader batch, bit-packed + verified on the device inside the step (input_form='
                                  "'tensor_representation')" if surface_struct else 'dense (2, N, N) fp32 tensor representation')),
                       'block1': ('structured: class tables + closed-form per-channel product (csrc/block1_struct.hip)'
                                  if ((struct1 and xbits is not None) or surface_struct) else 'generic'),
                       'mlp_contraction': ('v_mfma_f32_32x32x16_bf16' if bf16 else
                                           ('mlp1 / mlp2 (forward, and the pair backward fgnn_mlp_bwd_pair_x3): 3 x bf16 split operands (8 / 6 partial '
                                            'products, fp32 accumulation) on v_mfma_f32_32x32x16_bf16; mlp3: v_mfma_f32_32x32x2_f32' if getattr(eng, 'x3', False)
                                            else ('forward: v_mfma_f32_32x32x2_f32; backward (mlp1 + mlp2 pair, mlp3): v_mfma_f32_16x16x4_f32 on 16-pixel tiles (csrc/*_t16.hip; FGNN_T16=%s)'
                                                  % getattr(eng, 'T16', '?') if getattr(eng, 'T16', '0') not in ('0', '') else 'v_mfma_f32_32x32x2_f32'))),
                       'env_switches': {k: v for k, v in os.environ.items() if k.startswith('FGNN_') and k not in ('FGNN_BENCH_CHILD',)},
                       'grad_allreduce': ('%s sum of %d fp32 per step' % (torch.distributed.get_backend(), layout.total)) if world > 1 else 'none'},
            'ranks_seen': dp.world_size(), 'backend': torch.distributed.get_backend() if world > 1 else None, 'allreduce_ms': allreduce_ms,
            'roofline': roofline, 'step_trace_ms': step_trace,
            'step_model': {'algorithmic_gflop_per_pair': fl_pair / 1e9, 'algorithmic_mb_per_pair': by_pair / 1e6,
                           'hbm_frac_of_8TBs': value / world * by_pair / (HBM_PEAK_GBS * 1e9),
                           'mfma_frac_of_peak': value / world * fl_pair / ((MFMA_BF16_PEAK_TF if bf16 else MFMA_F32_PEAK_TF) * 1e12),
                           # the fractions above price the DENSE algorithm at this step time; with block 1 on its structured
                           # input part of that work is not executed -- these price what the step really runs
                           'executed_gflop_per_pair': fl_exec / 1e9, 'executed_mb_per_pair': by_exec / 1e6,
                           'hbm_frac_executed': value / world * by_exec / (HBM_PEAK_GBS * 1e9),
                           'mfma_frac_executed': value / world * fl_exec / ((MFMA_BF16_PEAK_TF if bf16 else MFMA_F32_PEAK_TF) * 1e12)},
            'kernels': kernels,
        }
        if cpu_leg:
            out['cpu_baseline'] = cpu_baseline_ragged(layout, params, xs, ys) if ragged else cpu_baseline(layout, params, x1, x2)
        else:
            out['cpu_baseline'] = None
        return out
    return None


if __name__ == '__main__':
    main()
