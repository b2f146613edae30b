"""GPU, 2 ranks: the data-parallel TPGSR / TSRN train steps themselves.  With two GPUs the ranks talk over RCCL; on a one-GPU box
both ranks share cuda:0 and talk over gloo (device tensors staged through the host by torch) -- everything of the multi-rank step
except RCCL itself runs on the HIP path: parameter broadcast, bucket launches inside the backward pass, the averaging kernel, clip + Adam.
"2 ranks x batch b == 1 rank accumulating 2 micro-batches of b with per-micro-batch BatchNorm statistics" (SURVEY 8e),
checked on the flat gradient buffer after the exchange, and identical parameters on both ranks after the step."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _build(dev):
    sys.path.insert(0, ROOT)
    from oracle import tpgsr_oracle as O
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, tps_hw=(16, 64)))
    teacher = crnn.CRNN(32, 1, 37, 256)
    teacher.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 12))
    stu = crnn.CRNN(32, 1, 37, 256)
    stu.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 13))
    return sr.to(dev).train(), stu.to(dev).train(), teacher.to(dev).eval()


def _worker(rank, world, port, q, shared_gpu):
    try:
        _worker_body(rank, world, port, q, shared_gpu)
    except Exception:       # hand the traceback to the parent instead of dying silently
        import traceback
        q.put((rank, traceback.format_exc(), None))


def _worker_body(rank, world, port, q, shared_gpu):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", 0 if shared_gpu else rank)
    torch.cuda.set_device(dev)
    if shared_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from oracle import tpgsr_oracle as O
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    sr, stu, teacher = _build(dev)
    ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1, world_size=world)
    ts.broadcast_parameters(0)
    lr, hr = O.synthetic_batch(4, 1234 + rank)
    ts.pool.bind(dev)
    teacher._engine().bind(dev)
    ts._phase_a(lr.to(dev), hr.to(dev))       # launches bucket 0 inside the backward pass
    ts._exchange()
    torch.cuda.synchronize()
    g = ts.pool.grad.clone().cpu()
    ts._phase_b()
    torch.cuda.synchronize()
    q.put((rank, g.numpy(), ts.pool.flat.clone().cpu().numpy()))      # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_tpgsr_step_equals_two_micro_batches():
    shared_gpu = torch.cuda.device_count() < 2
    sys.path.insert(0, ROOT)
    from oracle import tpgsr_oracle as O
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shared_gpu)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda t: t[0])
    for r in res:
        assert not isinstance(r[1], str), f"rank {r[0]} failed:\n{r[1]}"
    for pr in procs:
        pr.join(60)
    (_, g0, p0), (_, g1, p1) = [(r, torch.from_numpy(g), torch.from_numpy(p_)) for r, g, p_ in res]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    # single process: accumulate the two micro-batches (the arena accumulates like autograd), average
    dev = torch.device("cuda", 0)
    sr, stu, teacher = _build(dev)
    ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
    ts.pool.bind(dev)
    teacher._engine().bind(dev)
    acc = None
    for r in range(2):
        lr, hr = O.synthetic_batch(4, 1234 + r)
        ts._phase_a(lr.to(dev), hr.to(dev))   # zero_grad + fwd + bwd
        torch.cuda.synchronize()
        acc = ts.pool.grad.clone() if acc is None else acc + ts.pool.grad
    acc = (acc / 2).cpu()
    assert (acc - g0).abs().max() <= 2e-5 * acc.abs().max()
