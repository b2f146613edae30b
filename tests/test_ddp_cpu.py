"""CPU / gloo, world_size 2: the data-parallel exchange logic and its semantics:
two ranks x batch b with summed-then-averaged flat gradients == one rank accumulating two micro-batches of b with
per-micro-batch BatchNorm statistics, and identical parameters on every rank after the step."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _flat(grads):
    return torch.cat([g.reshape(-1) for g in grads])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import tpgsr_oracle as O
    from tpgsr_amd.distributed import broadcast_state, exchange_gradients, shard_seed
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, srb_nums=1), 5 + rank)   # ranks start DIFFERENT on purpose
    p = O.as_params(sd)
    keys = O.trainable_keys(p)
    flat = _flat([p[k].detach() for k in keys])
    bufs = [v for k, v in p.items() if "running_" in k]
    broadcast_state(flat, bufs, 0)                                  # rank 0's parameters everywhere
    ofs = 0
    with torch.no_grad():
        for k in keys:
            n = p[k].numel()
            p[k].copy_(flat[ofs:ofs + n].view_as(p[k]))
            ofs += n
    lr, hr = O.synthetic_batch(2, shard_seed(1234, rank))
    sr = O.tsrn_forward(p, lr, training=True, stn=False, srb_nums=1)
    loss = O.image_loss(sr, hr).mean() * 100
    grads = torch.autograd.grad(loss, [p[k] for k in keys])
    g = _flat(grads)
    exchange_gradients(g)                                            # ONE flat all-reduce
    g /= world
    opt = O.AdamState([p[k] for k in keys])
    gl, ofs = [], 0
    for k in keys:
        n = p[k].numel()
        gl.append(g[ofs:ofs + n].view_as(p[k]).clone())
        ofs += n
    O.clip_grad_norm_(gl, 0.25)
    opt.step(gl)
    q.put((rank, _flat([p[k].detach() for k in keys]), g))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_exchange_matches_gradient_accumulation():
    sys.path.insert(0, ROOT)
    from oracle import tpgsr_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    (_, p0, g0), (_, p1, g1) = res
    assert torch.equal(p0, p1), "ranks diverged after an identical update"
    assert torch.equal(g0, g1)
    # single-process reference: two micro-batches (BN statistics per micro-batch), gradients averaged
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, srb_nums=1), 5)
    p = O.as_params(sd)
    keys = O.trainable_keys(p)
    acc = None
    for r in range(2):
        lr, hr = O.synthetic_batch(2, 1234 + r)
        sr = O.tsrn_forward(p, lr, training=True, stn=False, srb_nums=1)
        loss = O.image_loss(sr, hr).mean() * 100
        gr = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, [p[k] for k in keys])])
        acc = gr if acc is None else acc + gr
    acc /= 2
    assert (acc - g0).abs().max() <= 1e-5 * acc.abs().max()


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tpgsr_amd.distributed import GradientExchanger
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1000, generator=g)
    mine = flat.clone()
    ex = GradientExchanger(flat, [(0, 384), (384, 1000)], None)
    ex.launch(0)                       # the SR-net bucket leaves while "the student backward" still writes bucket 1
    flat[384:] += 1.0
    try:
        ex.launch(0)
        twice = False
    except RuntimeError:
        twice = True
    ex.finish()                        # launches bucket 1, waits for both, averages
    ex.launch(0)                       # next step: buckets can be launched again
    ex.finish()
    q.put((rank, mine, flat.clone(), twice))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_overlapped_exchange_two_ranks():
    """GradientExchanger (the host logic of TPGSRTrainStep._exchange): early launch of bucket 0, late bucket 1, average,
    re-usable every step; both ranks end with identical buffers = the mean of their inputs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    (_, a0, f0, tw0), (_, a1, f1, tw1) = res
    assert tw0 and tw1, "launching a bucket twice in one step must raise"
    want = (a0 + a1) / 2
    want[384:] += 1.0
    # second exchange of already identical buffers is the identity
    assert torch.equal(f0, f1)
    assert (f0 - want).abs().max() < 1e-6


def _three_bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tpgsr_amd.distributed import GradientExchanger
    g = torch.Generator().manual_seed(200 + rank)
    flat = torch.randn(1000, generator=g)
    mine = flat.clone()
    # the C3 step's layout: SR networks [0, 300) | gap + the generator's first layers [300, 340) | the generator from conv3 on [340, 1000):
    # bucket 1 is the LATER address range and leaves first (between the generator's two backward plans), bucket 2 at the end
    ex = GradientExchanger(flat, [(0, 300), (340, 1000), (300, 340)], None)
    ex.begin()
    ex.launch(0)
    flat[300:] += 1.0                  # "the generator's backward" still writes both of its ranges
    ex.launch(1)
    flat[300:340] += 2.0               # ... and only the first layers after the early bucket left
    ex.finish()
    q.put((rank, mine, flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_buckets_launched_out_of_address_order_two_ranks():
    """the bounds TPGSRTrainStep hands the exchanger with one text-prior generator: three buckets, the middle address range last"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_three_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    (_, a0, f0), (_, a1, f1) = res
    want = (a0 + a1) / 2
    want[300:] += 1.0
    want[300:340] += 2.0
    assert torch.equal(f0, f1)
    assert (f0 - want).abs().max() < 1e-6


def _seven_bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tpgsr_amd.distributed import GradientExchanger
    g = torch.Generator().manual_seed(300 + rank)
    flat = torch.randn(2200, generator=g)
    mine = flat.clone()
    # the C5 step's layout (three text-prior generators behind the SR networks): per student an EARLY bucket = the generator from
    # conv3 on (its first backward plan) and a REST bucket = the gap + its first layers (second plan).  The step runs the students'
    # backwards last-to-first, so the launch order is 5, 6, 3, 4, 0 (SR networks: their final stage ends later), 1 and -- at finish -- 2.
    bounds = [(0, 300), (340, 900), (300, 340), (940, 1500), (900, 940), (1540, 2200), (1500, 1540)]
    ex = GradientExchanger(flat, bounds, None)
    ex.begin()
    written = torch.zeros(2200)
    left = set()
    for step, b in enumerate([5, 6, 3, 4, 0, 1]):
        ex.launch(b)
        left.add(b)
        # "the backward passes still running" write every range that has not left yet
        for j, (lo, hi) in enumerate(bounds):
            if j not in left:
                flat[lo:hi] += float(step + 1)
                written[lo:hi] += float(step + 1)
    ex.finish()
    q.put((rank, mine, flat.clone(), written))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_seven_buckets_of_the_three_student_step_two_ranks():
    """the bounds TPGSRTrainStep hands the exchanger with three text-prior generators (C5; tests/test_plan_dryrun_cpu.py pins that
    order against the recorded plans): every student its own early bucket, all seven launched out of address order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_seven_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    (_, a0, f0, w0), (_, a1, f1, w1) = res
    assert torch.equal(w0, w1)
    want = (a0 + a1) / 2 + w0
    assert torch.equal(f0, f1)
    assert (f0 - want).abs().max() < 1e-5


def test_arena_pool_and_dataparallel_wrapper_cpu():
    """ArenaPool layout (contiguous, 256-byte aligned slices, SR nets before students) and the .module-exposing wrapper's
    state_dict prefix -- host logic only, no kernels (the engines bind lazily on the GPU)."""
    sys.path.insert(0, ROOT)
    from tpgsr_amd.distributed import DataParallel
    from tpgsr_amd.engine import ArenaPool, ParamArena
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    stu = crnn.CRNN(32, 1, 37, 256)
    pool = ArenaPool([sr, stu, sr])                     # duplicates collapse (sr_share)
    assert len(pool.modules) == 2
    n_sr = ParamArena(sr).layout()[1]
    n_stu = ParamArena(stu).layout()[1]
    assert n_sr >= sum(p.numel() for p in sr.parameters()) and n_sr % 4 == 0
    assert n_stu >= sum(p.numel() for p in stu.parameters())
    dp = DataParallel(sr)
    assert dp.module is sr
    keys = list(dp.state_dict().keys())
    assert keys and all(k.startswith("module.") for k in keys)
    assert [k[len("module."):] for k in keys] == list(sr.state_dict().keys())   # save_checkpoint's netG.module.state_dict()


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tpgsr_amd.distributed import DataParallel
    torch.manual_seed(10 + rank)                                   # ranks start DIFFERENT on purpose
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Tanh(), torch.nn.Linear(3, 2))
    dp = DataParallel(net)                                          # no fused engine: per-parameter hooks + rank 0's weights
    p0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    x = torch.randn(5, 4, generator=torch.Generator().manual_seed(100 + rank))
    dp(x).pow(2).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()

    # the fused path's hook (what the network's single autograd node calls at the end of its backward pass) on a stand-in engine
    class _Arena:
        pass

    class _Eng:
        FUSED = True
        arena = _Arena()

        def bind(self, dev):
            pass

    eng = _Eng()
    eng.arena.flat = torch.full((6,), float(rank))
    eng.arena.grad = torch.arange(6, dtype=torch.float32) * (rank + 1)

    class _Fused(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def _engine(self):
            return eng

    fused = _Fused()
    dpf = DataParallel(fused)
    assert fused._grad_sync == dpf._sync
    fused._grad_sync(eng)
    q.put((rank, p0, g, eng.arena.flat.clone(), eng.arena.grad.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dataparallel_wrapper_two_ranks_hooked_and_fused_paths():
    """tpgsr_amd.distributed.DataParallel with world size 2: an operator-by-operator module (no fused engine) gets rank 0's weights
    and per-parameter gradient averaging; the fused path's _grad_sync averages the engine's flat gradient arena."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    (_, pa, ga, fa, gfa), (_, pb, gb, fb, gfb) = res
    assert torch.equal(pa, pb), "broadcast of rank 0's parameters"
    assert torch.equal(ga, gb), "ranks must hold the same averaged gradient"
    # single process: mean of the two ranks' gradients with rank 0's weights
    torch.manual_seed(10)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Tanh(), torch.nn.Linear(3, 2))
    acc = None
    for r in range(2):
        net.zero_grad()
        x = torch.randn(5, 4, generator=torch.Generator().manual_seed(100 + r))
        net(x).pow(2).sum().backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        acc = g.clone() if acc is None else acc + g
    assert (acc / 2 - ga).abs().max() < 1e-6
    assert torch.equal(fa, torch.zeros(6)) and torch.equal(fb, torch.zeros(6))          # flat parameters: rank 0's
    assert torch.equal(gfa, gfb) and (gfa - torch.arange(6, dtype=torch.float32) * 1.5).abs().max() < 1e-6
