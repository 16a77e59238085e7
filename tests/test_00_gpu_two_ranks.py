"""GPU: the HIP trainer with MORE THAN ONE RANK.  Two fresh child processes (rendezvous on 127.0.0.1; one GPU per rank
over RCCL when the node has two GPUs, else both on cuda:0 over gloo -- graph_neural_net_amd.dp.pick_backend) run
FgnnTrainer.train_step -- eager, HIP-graph-captured, size-bucketed ragged and padded ragged -- on their shards
of a global batch; the parameters after three optimizer steps must equal the single-process run on the concatenated batch
("equals the single-process batch"; normaliser toolbox/losses.py:27-34), and every step must issue
exactly ONE collective.

The children are started from a process that has not touched the GPU (this file sorts first and checks it); a process that
has initialised the GPU is never replaced by another program.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dp_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, out, tmp_path):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'))
        log = open(os.path.join(tmp_path, 'w%d_r%d.log' % (world, rank)), 'w')
        procs.append((subprocess.Popen([sys.executable, WORKER, out], env=env, stdout=log, stderr=subprocess.STDOUT), log))
    for p, log in procs:
        try:
            rc = p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        log.close()
        assert rc == 0, open(log.name).read()[-3000:]


def test_two_ranks_equal_single_process(tmp_path):
    if torch.cuda.is_initialized():
        pytest.skip('the GPU is already initialised in this process; run this file first / on its own')
    tmp = str(tmp_path)
    _launch(1, os.path.join(tmp, 'single'), tmp)
    _launch(2, os.path.join(tmp, 'double'), tmp)
    one = torch.load(os.path.join(tmp, 'single.pt'))
    two = torch.load(os.path.join(tmp, 'double.pt'))
    p0 = one['p0']
    for mode in ('eager', 'capture', 'ragged', 'padded'):
        # (1) the sharp check: after the FIRST step's all-reduce both world sizes hold the same buffer -- the gradient sum of
        #     the global batch (shard sums differ from the full-batch sum only by summation order: <= 2.4e-6 relative,
        #     SURVEY.md 8c), the global loss sum and the global node count
        c1, c2 = one[mode + '_comm'], two[mode + '_comm']
        assert c1[-1] == c2[-1] and c1[-1] > 0                                   # node count: exact
        assert abs(c1[-2] - c2[-2]) < 1e-5 * abs(c1[-2])                         # loss sum
        assert ((c1[:-2] - c2[:-2]).norm() / c1[:-2].norm()).item() < 1e-5, mode
        p1, l1, t1 = one[mode]
        p2, l2, t2 = two[mode]
        assert t1 == t2 == 3
        assert abs(l1[0] - l2[0]) < 1e-5 * abs(l1[0])                            # the loss of the GLOBAL batch on every rank
        # (2) after three Adam steps: Adam turns the noise-level entries of the gradient (the analytically zero last-conv
        #     biases) into +-lr moves in a noise-determined direction in BOTH runs, so later
        #     losses / parameters agree only to that level: compare the parameter MOVEMENT
        for a, b in zip(l1, l2):
            assert abs(a - b) < 1e-3 * abs(a), (mode, l1, l2)
        assert ((p1 - p2).norm() / (p1 - p0).norm()).item() < 0.1, (mode, ((p1 - p2).norm() / (p1 - p0).norm()).item())
    # captured == eager (same kernels, same order): bit-identical parameters in both world sizes
    assert torch.equal(one['eager'][0], one['capture'][0])
    assert torch.equal(two['eager'][0], two['capture'][0])


def test_one_collective_per_step(tmp_path):
    """Runtime count: inside each child every data-moving entry point of torch.distributed is wrapped by a counter
    (tests/dp_worker.py), and each FgnnTrainer step -- eager, captured (including its capturing first step), size-bucketed
    ragged, padded ragged -- must have issued exactly ONE collective, an all_reduce (north_star: "a single RCCL all-reduce of
    gradients ... per step"; the loss sum and the node count ride in the same buffer)."""
    if torch.cuda.is_initialized():
        pytest.skip('the GPU is already initialised in this process; run this file first / on its own')
    tmp = str(tmp_path)
    _launch(2, os.path.join(tmp, 'count'), tmp)
    two = torch.load(os.path.join(tmp, 'count.pt'))
    assert two['backend'] == ('nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    for mode in ('eager', 'capture', 'ragged', 'padded'):
        per_step = two[mode + '_collectives']
        assert len(per_step) == 3
        if mode == 'capture' and two['capture_in_graph']:
            # RCCL: the collective was recorded into the step's graph -- ONE call while capturing, none from the host after
            # (+ one flag all-reduce while capturing: the ranks agree that the capture worked on all of them, trainer.py)
            assert per_step == [['all_reduce', 'all_reduce'], [], []], per_step
            continue
        for calls in per_step:
            assert calls == ['all_reduce'], (mode, per_step)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent (which never touches the GPU) starts two rank processes,
    relays rank 0's JSON line and exits 0.  On a node with two GPUs this IS the 2-GPU scaling run (default backend nccl =
    RCCL, the all-reduce inside the replayed graph); on a one-GPU box both ranks share cuda:0 over gloo."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    multi = torch.cuda.device_count() >= 2          # (device_count does not initialise the GPU)
    backend = 'nccl' if multi else 'gloo'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--windows', '2',
                        '--settle', '4', '--no-cpu-baseline', '--profile-steps', '0'] + ([] if multi else ['--backend', 'gloo']),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['backend'] == backend
    assert out['config']['allreduce_in_graph'] == multi
    assert out['config']['global_batch'] == 64 and out['config']['parallelism'] == 'dp2' and out['scaling'] == 'weak'
    assert out['allreduce_ms'] is not None and out['allreduce_ms'] > 0
    assert out['value'] > 0 and abs(out['value'] - 64 / (out['ms_per_step'] * 1e-3)) < 1e-6 * out['value']
    assert out['ms_per_step_min'] <= out['ms_per_step'] <= out['ms_per_step_max'] and out['timed_windows'] == 2


@pytest.mark.parametrize('config', ['cfg2', 'cfg5'])
def test_bench_eight_rank_dress_rehearsal(tmp_path, config):
    """The exact commands of the 8-GPU scaling run (`bench.py --gpus 8`, `--config cfg5 --gpus 8`: BASELINE configs 3 and 5) in the
    functional form a one-GPU box can host -- eight fresh rank processes sharing cuda:0 over gloo (one GPU per rank over RCCL when
    the node has eight): all eight ranks are seen, the global batch is 256 / 64 pairs, every timed step of every rank issues exactly
    ONE collective (an all_reduce), every rank ends with the same buffer, and that buffer -- the gradient of the global batch, with
    the global node count as normaliser -- equals a single-process run on the concatenated batch."""
    import json
    if torch.cuda.is_initialized():
        pytest.skip('the GPU is already initialised in this process; run this file first / on its own')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    multi = torch.cuda.device_count() >= 8
    vdir = os.path.join(str(tmp_path), 'verify')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--windows', '1', '--settle', '2',
           '--no-cpu-baseline', '--profile-steps', '0', '--verify-dir', vdir] + ([] if multi else ['--backend', 'gloo']) \
        + (['--config', 'cfg5'] if config == 'cfg5' else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    pairs = 8 if config == 'cfg5' else 32
    assert out['n_gpus'] == 8 and out['ranks_seen'] == 8 and out['backend'] == ('nccl' if multi else 'gloo')
    assert out['config']['global_batch'] == 8 * pairs and out['config']['parallelism'] == 'dp8' and out['scaling'] == 'weak'
    assert abs(out['value'] - 8 * pairs / (out['ms_per_step'] * 1e-3)) < 1e-6 * out['value']
    ranks = [torch.load(os.path.join(vdir, 'rank%d.pt' % k)) for k in range(8)]
    for k, d in enumerate(ranks):
        assert d['rank'] == k and d['world'] == 8 and d['batch_per_gpu'] == pairs
        per_step = d['collectives_per_step']
        assert len(per_step) == 2
        if d['allreduce_in_graph']:         # RCCL: recorded into the replayed graph -- no host call per step
            assert per_step == [[], []], (k, per_step)
        else:
            assert per_step == [['all_reduce'], ['all_reduce']], (k, per_step)
        assert torch.equal(d['comm'], ranks[0]['comm']), k                 # one reduced buffer on every rank
        assert d['total_nodes'] == ranks[0]['total_nodes']
    if config == 'cfg5':                    # the ranks hold different node counts; the normaliser is their sum
        assert len({tuple(d['sizes']) for d in ranks}) == 8
        assert ranks[0]['total_nodes'] == float(sum(sum(d['sizes']) for d in ranks))
    # the single-process run on the concatenated batch (a fresh child: this process never touches the GPU)
    ref_path = os.path.join(str(tmp_path), 'ref.pt')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_bench_reference.py'), config, '8', ref_path], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = torch.load(ref_path)
    got = ranks[0]['comm'].double()
    loss = sum(d['loss_local'] for d in ranks)
    assert ref['total_nodes'] == ranks[0]['total_nodes']
    l2 = lambda a, b: ((a - b.double()).norm() / b.double().norm()).item()
    assert abs(loss - ref['concat_loss']) < 1e-5 * abs(ref['concat_loss'])
    if config == 'cfg2':
        # same kernels, same tile geometry per graph; the weight-gradient partials are summed in another grouping (8 ranks x
        # workgroups, then the all-reduce): summation-order differences only (<= 2.4e-6 relative)
        assert l2(got, ref['concat']) < 1e-5, l2(got, ref['concat'])
    else:
        # the shards on their own padded geometry, summed on one device: what the ranks computed, up to the order of the 8-term sum
        assert l2(got, ref['shards']) < 1e-6, l2(got, ref['shards'])
        # the ONE concatenated batch pads every graph to the global largest one: other tile boundaries, so the GraphNorm statistics
        # are summed in another order and the gradients agree within the rounding / near-tie class of a 4-block N <= 120 step
        assert l2(got, ref['concat']) < 5e-3, l2(got, ref['concat'])


def test_bench_stops_all_ranks_when_one_dies_at_start_up(tmp_path):
    """A rank that dies at start-up (bad device index, RCCL init failure ...) must not leave the others waiting in the rendezvous
    until the distributed timeout: bench.py's launcher polls all of its rank processes, stops the others and exits non-zero at
    once.  FGNN_BENCH_FAIL_RANK=3 makes rank 3 exit with code 3 at the start of main()."""
    import time
    if torch.cuda.is_initialized():
        pytest.skip('the GPU is already initialised in this process; run this file first / on its own')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', FGNN_BENCH_FAIL_RANK='3')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--backend', 'gloo',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    dt = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert 'rank 3 exited with code 3' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]            # no result line from a broken job
    assert dt < 120, dt                      # seconds (process start-up of the image), not the 10-minute rendezvous timeout


def test_rccl_all_reduce_on_the_gradient_buffer(tmp_path):
    """The backend the N-GPU run uses (nccl = RCCL) initialises on this box and sums the trainer's communication buffer
    [flat gradient | loss sum | node count] in place on the device -- with the one rank a one-GPU box can host (the two-rank
    tests above share cuda:0, which RCCL refuses, and therefore run on gloo).  A fresh child process: this one never touches
    the GPU before starting it."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "flat = torch.randn(40002, device='cuda:0'); ref = flat.clone()\n"
        "dist.all_reduce(flat, op=dist.ReduceOp.SUM); torch.cuda.synchronize()\n"
        "assert torch.equal(flat, ref)\n"
        "e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)\n"
        "e0.record()\n"
        "for _ in range(20): dist.all_reduce(flat, op=dist.ReduceOp.SUM)\n"
        "e1.record(); torch.cuda.synchronize()\n"
        "print('RCCL_OK backend=%s us_per_call=%.1f' % (dist.get_backend(), e0.elapsed_time(e1) * 50))\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_OK backend=nccl' in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_rccl_all_reduce_inside_the_captured_step(tmp_path):
    """The RCCL all-reduce recorded INTO the trainer's HIP graph (model work -> all-reduce -> fused Adam = one replay), with the
    one rank a one-GPU box can host (collective='always' issues the collective although there is nothing to sum): three
    replayed steps must leave bit-identical parameters to the trainer without a collective, the host must have issued the
    collective only while capturing (the recorded all-reduce + one flag all-reduce by which the ranks agree that the capture worked
    everywhere), and replays must not call into torch.distributed at all."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from graph_neural_net_amd import dp, synthetic\n"
        "from graph_neural_net_amd.engine import ParamLayout\n"
        "from graph_neural_net_amd.trainer import FgnnTrainer\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "calls = []\n"
        "orig = dist.all_reduce\n"
        "dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]\n"
        "dev = torch.device('cuda:0')\n"
        "lay = ParamLayout(2, 2, 32, 32, 3)\n"
        "p0 = lay.init_flat(11, dev)\n"
        "batches = [synthetic.make_batch(7100 + s, 4, 18, 'ErdosRenyi', 0.3, 0.05) for s in range(3)]\n"
        "out = []\n"
        "for coll in ('auto', 'always'):\n"
        "    tr = FgnnTrainer(lay, p0.clone(), lr=2e-3, capture=True, collective=coll)\n"
        "    n0 = len(calls); per = []; losses = []\n"
        "    for x1, x2 in batches:\n"
        "        m0 = len(calls)\n"
        "        loss, _ = tr.train_step(x1.to(dev), x2.to(dev))\n"
        "        losses.append(loss.item()); per.append(len(calls) - m0)\n"
        "    torch.cuda.synchronize()\n"
        "    out.append((tr.params.clone(), losses, per, tr.allreduce_in_graph))\n"
        "(pa, la, ca, ga), (pb, lb, cb, gb) = out\n"
        "assert not ga and ca == [0, 0, 0], (ga, ca)\n"
        "assert gb and cb == [2, 0, 0], (gb, cb)      # the captured all-reduce + the ranks' agreement flag, once per shape\n"
        "assert torch.equal(pa, pb) and la == lb, (la, lb)\n"
        "print('RCCL_IN_GRAPH_OK')\n"
        "dist.destroy_process_group()\n") % (ROOT,)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_IN_GRAPH_OK' in r.stdout, (r.stdout + r.stderr)[-3000:]
