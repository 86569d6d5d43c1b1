"""GPU: the direct RCCL collective behind the C ABI (aph_comm_* / aph_allreduce_f32).
One-rank communicator on any box (the all-reduce is then the identity, but it is a real RCCL launch: eager, and as a node
of the step's hipGraph next to a mid-run device synchronize -- the pattern that broke the torch.distributed + graph-replay
combination in round 1); two ranks when the box has two GPUs (skipped on the 1-GPU boxes)."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def test_one_rank_allreduce_and_error_convention():
    from aphantasia_amd import comm as acomm
    uid = acomm.new_unique_id()
    assert len(uid) == 128 and any(uid)
    c = acomm.Comm(0, 1, uid)
    x = torch.randn(2769120, device=DEV)
    y = x.clone()
    c.all_reduce_(y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(ValueError):
        c.all_reduce_(x.half())
    with pytest.raises(RuntimeError):
        acomm.Comm(3, 2, uid)                  # rank outside 0..nranks-1: APH_ERR_ARG, no RCCL call
    c.close()


def test_step_with_in_graph_allreduce_matches_plain_step():
    """whole step (gradient + RCCL all-reduce + Adam) captured as ONE hipGraph, a device synchronize in the middle of the run:
    bit-identical to the engine without a communicator"""
    from aphantasia_amd import clip as aclip, comm as acomm, transforms
    from aphantasia_amd.engine import Engine
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=24)
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    c = acomm.Comm(0, 1, acomm.new_unique_id())

    def run(comm, graph):
        seed_all(0)
        h, w = 360, 640
        params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
        eng = Engine(params, h, w, model, 24, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph=graph,
                     comm=comm, reduce_always=comm is not None)
        losses = []
        for i in range(8):
            losses.append(eng.step())
            if i == 4:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        run.last = eng
        return eng.params.clone(), eng.grad.clone(), [float(l) for l in losses[-1:]]
    a = run(None, False)
    b = run(c, False)
    d = run(c, True)
    # the capture self-check (one eager step against one replay from the same state, agreed through the communicator) ran and passed
    assert run.last._graph_checked and run.last.use_graph and run.last._graphs is not None
    for x, y in ((a, b), (a, d)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and x[2] == y[2]
    assert torch.isfinite(a[0]).all()
    # a self-check that FAILS: the engine restores the state it started the check from, stays eager, and the run is still the same run
    saved = Engine._same_bits
    Engine._same_bits = staticmethod(lambda p, q: False)
    try:
        e = run(c, True)
    finally:
        Engine._same_bits = saved
    assert run.last._graph_checked and not run.last.use_graph and run.last._graphs is None
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1]) and a[2] == e[2]


def _rank_main(rank, world, port, ret, same_device=False, graph_allreduce=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), APH_RUN_ID='t%d' % port, HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0 if same_device else rank)
    from aphantasia_amd import clip as aclip, comm as acomm, transforms
    from aphantasia_amd.engine import Engine
    try:
        c = acomm.create(rank, world, key='t%d' % port) if world > 1 else None
    except Exception as e:                      # (RCCL's refusal of two ranks on one device arrives here)
        ret[rank] = ('comm-failed', repr(e))
        return
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=24)
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    seed_all(0)
    h, w = 360, 640
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
    eng = Engine(params, h, w, model, 24, [(target, -1.0)], sim='mix', transform=transforms.normalize(), macro=0.4, rank=rank, world=world, comm=c,
                 rng='reference', graph_allreduce=graph_allreduce)
    losses = []
    for i in range(6):
        seed_all(100 + i)
        eng.step()
        losses.append(eng.global_loss())
    torch.cuda.synchronize()
    ret[rank] = (eng.params.cpu(), losses, bool(eng.use_graph and eng._graphs is not None))


def _two_ranks(same_device, graph_allreduce, port, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    ret = mgr.dict()
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, ret, same_device, graph_allreduce)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(timeout)
    hung = [p for p in ps if p.is_alive()]
    for p in hung:
        p.kill()                                # (exact processes this test started)
    return dict(ret), bool(hung), ctx, mgr


@pytest.mark.parametrize('graph_allreduce', [False, True])
def test_two_ranks_on_one_device_if_rccl_permits(graph_allreduce):
    """a REAL 2-rank RCCL communicator on a one-GPU box -- two processes on cuda:0 -- with the step's all-reduce eager and
    as a node of the step's hipGraph.  RCCL (like NCCL) rejects two ranks of one communicator on the same device ("Duplicate GPU detected"):
    when it does, the test is skipped WITH the library's own message, so the log says what was tried and why it could not run here."""
    from aphantasia_amd.comm import free_port
    ret, hung, ctx, mgr = _two_ranks(True, graph_allreduce, free_port(), timeout=180)
    refused = [v for v in ret.values() if isinstance(v, tuple) and v and v[0] == 'comm-failed']
    if refused or hung or len(ret) < 2:
        pytest.skip('RCCL does not form a 2-rank communicator on one device: %s' % (refused[0][1][:300] if refused else ('ranks hung in ncclCommInitRank (killed after 180 s)' if hung else 'a rank died')))
    assert torch.equal(ret[0][0], ret[1][0]) and ret[0][1] == ret[1][1]
    assert ret[0][2] == ret[1][2] == graph_allreduce


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
@pytest.mark.parametrize('graph_allreduce', [False, True])
def test_two_ranks_bit_identical_params_and_loss_curve(graph_allreduce):
    """R = 2 over RCCL: parameters bit-identical across ranks after every step's replicated Adam, loss curve equal to R = 1 to
    fp32 rounding; eager all-reduce and (opt-in) the all-reduce as a node of the step's hipGraph"""
    ret, hung, ctx, mgr = _two_ranks(False, graph_allreduce, 29731 + int(graph_allreduce))
    assert not hung and len(ret) == 2
    assert ret[0][2] == ret[1][2] == graph_allreduce          # the captured multi-rank step passed its self-check on both ranks (or eager was asked for)
    single = mgr.dict()
    p = ctx.Process(target=_rank_main, args=(0, 1, 29733, single)); p.start(); p.join(600)
    assert torch.equal(ret[0][0], ret[1][0])
    assert ret[0][1] == ret[1][1]
    assert np.abs(np.array(ret[0][1]) - np.array(single[0][1])).max() < 2e-5
