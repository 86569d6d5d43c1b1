"""CPU: host logic of the fused step (aphantasia_amd/engine.py) -- RNG-order-exact crop draws, the
full train(i) chain against the oracle's free-running reference loop, and the world_size-2 sharded
path over gloo.  The kernels run under the tests/emu interpreter (test infrastructure); the product
engine runs the identical code on libaphantasia_hip.so (tests/test_gpu_step.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from aphantasia_amd import _ffi
from aphantasia_amd.engine import Engine, shard_range
from aphantasia_amd.utils import draw_crop_params
from aphantasia_amd.weights import synthetic_visual_weights
from oracle import reference_path as R
from oracle import clip_vit_ref

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))
TINY = dict(input_resolution=32, patch_size=16, width=256, layers=2, heads=4, output_dim=128)
H, W, S = 40, 56, 5


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


@pytest.mark.parametrize('align', ['uniform', 'central', 'overscan', 'overmax'])
def test_draws_match_reference_order(align):
    # oracle.draw_crop_table is pinned bit-exactly to the reference's slice_imgs (test_oracle.py)
    for seed in range(6):
        for (h, w, size, macro) in [(720, 1280, 224, 0.4), (48, 80, 16, 0.4), (360, 640, 224, 0.0), (300, 200, 224, 1.0)]:
            seed_all(seed)
            want = R.draw_crop_table(24, size, h, w, align, macro)
            tail_want = torch.rand(3)
            seed_all(seed)
            got, augs = draw_crop_params(24, size, h, w, align, macro)
            tail_got = torch.rand(3)
            assert augs is None and np.array_equal(want, got), (seed, h, w)
            assert torch.equal(tail_want, tail_got)      # consumed exactly the same number of draws


def test_fast_transform_draw_order():
    from aphantasia_amd.transforms import transforms_fast
    from oracle import augment_ref
    seed_all(11)
    want_aug = []
    want = R.draw_crop_table(12, 224, 720, 1280, 'uniform', 0.4, per_cut_hook=lambda c: want_aug.append(augment_ref.draw_fast_params(224)))
    seed_all(11)
    got, augs = draw_crop_params(12, 224, 720, 1280, 'uniform', 0.4, transforms_fast)
    assert np.array_equal(want, got)
    for a, b in zip(want_aug, augs):
        assert a['angle'] == b['angle'] and a['erase'] == b['erase']
        assert (a['persp'] is None) == (b['persp'] is None)
        if a['persp'] is not None:
            assert np.allclose(a['persp'], b['persp'], rtol=1e-6, atol=1e-7)


def test_shard_range_covers():
    for S_ in (190, 43, 7, 1):
        for world in (1, 2, 4, 8):
            spans = [shard_range(S_, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == S_
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _emu_lib():
    import build_emu
    return _ffi.Library(build_emu.build())


def _make_engine(lib, rank=0, world=1, pg=None, sim='mix', optimizer='adam_custom', S=S):
    from aphantasia_amd.clip import CLIPModel
    w = synthetic_visual_weights(TINY, 3)
    model = CLIPModel('tiny', TINY, w, None, max_batch=S, lib=lib)
    seed_all(0)
    params = R.fft_params_init([1, 3, H, W]).contiguous()
    target = torch.randn(1, 128, generator=torch.Generator().manual_seed(2))
    eng = Engine(params, H, W, model, S, [(target, -1.0)], sim=sim, macro=0.4, rank=rank, world=world,
                 process_group=pg, lib=lib, optimizer=optimizer)
    return eng, w, target


def test_engine_free_running_vs_oracle():
    lib = _emu_lib()
    eng, w, target = _make_engine(lib)
    run = R.ReferenceRun(H, W, lambda x: clip_vit_ref.encode_image(w, x, TINY), [(target, 1.0)], size=32,
                         params=eng.params.clone())
    seed_all(123)
    tables = [R.draw_crop_table(S, 32, H, W, 'uniform', 0.4) for _ in range(4)]
    for i, tb in enumerate(tables):
        want = run.step(tb)
        got = float(eng.step(tb))
        assert abs(got - want) < 1e-3, (i, got, want)
    with torch.no_grad():
        img = run.image(1.1)[0]
    assert (eng.synthesize(1.1) - img).pow(2).mean().sqrt().item() < 2e-2


def _worker(rank, world, port, q, cuts=S):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = _emu_lib()
    eng, w, target = _make_engine(lib, rank, world, S=cuts)
    seed_all(123)
    losses = []
    for _ in range(2):
        eng.step()
        losses.append(eng.global_loss())
    q.put((rank, eng.params.clone().numpy(), losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,cuts', [(2, 5), (3, 7)])
def test_engine_ranks_gloo(world, cuts):
    """SURVEY section 8(e) acceptance at small scale: the cuts split over `world` ranks (shards of unequal size: 3 + 2, 3 + 2 + 2), one
    all-reduce of the partial spectrum gradient per step -> every rank holds bit-identical parameters, the summed partial losses equal
    the single-rank loss, and the run equals the single-rank run up to summation order"""
    import torch.multiprocessing as mp
    _emu_lib()                                 # build once before forking workers
    lib = _emu_lib()
    eng, _, _ = _make_engine(lib, S=cuts)
    seed_all(123)
    want_losses = []
    for _ in range(2):
        eng.step()
        want_losses.append(eng.global_loss())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    from aphantasia_amd.comm import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cuts)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1])                  # ranks stay bit-identical
        assert res[0][2] == res[r][2]                                # and report the same global loss
    # == the single-rank run up to fp32 summation order of the partial gradients; Adam(beta1=0) turns
    # rounding-level differences of near-zero gradient entries into visible (but tiny) parameter differences
    d = np.abs(res[0][1] - eng.params.numpy())
    assert d.max() < 5e-3 and d.mean() < 5e-5, (d.max(), d.mean())
    assert np.allclose(res[0][2], want_losses, atol=1e-4)


def test_skip_count_bookkeeping_across_reset_params():
    """the guarded Adam counts overflowed (skipped) steps on the device; the host folds new skips into the bias-correction step
    counter every GUARD_EVERY calls -- but only the skips of the CURRENT optimiser instance: reset_params (illustrip's per-frame fresh Adam)
    snapshots the counter as a floor.  Faked here by writing the counter directly (CPU path of _check_overflow / reset_params):
    a skip BEFORE a reset is not charged to the new frame, a skip AFTER it is, two resets inside one look-up window lose nothing."""
    lib = _emu_lib()
    eng, _, _ = _make_engine(lib)
    E = eng.GUARD_EVERY
    p0 = eng.params.clone()

    def look(calls_multiple=1):
        eng._calls = E * calls_multiple           # the counter is looked at when _calls is a multiple of GUARD_EVERY
        eng._check_overflow()

    scale0 = eng.loss_scale
    # (1) one skip in frame A, then the reset: the new frame's optimiser must not pay for it
    eng.guard[0] = 1
    eng.reset_params(p0)
    assert eng._state['step'][0] == 0 and eng._guard_floor == 1
    eng._state['step'][0] = 3                     # three steps of frame B taken
    look()
    assert eng._state['step'][0] == 3 and eng._guard_seen == 1
    assert eng.loss_scale == scale0 * 0.5 and eng._graphs is None       # (the overflow is still answered with a smaller loss scale)
    # (2) a genuine skip of frame B right after the reset IS charged
    eng.guard[0] = 2
    look(2)
    assert eng._state['step'][0] == 2 and eng._guard_seen == 2
    # (3) two resets inside one window: skip in frame B (counter 3), reset -> frame C, skip in frame C (counter 4), reset -> frame D
    eng.guard[0] = 3
    eng.reset_params(p0)
    eng.guard[0] = 4
    eng.reset_params(p0)
    assert eng._guard_floor == 4
    eng._state['step'][0] = 5
    look(3)
    assert eng._state['step'][0] == 5 and eng._guard_seen == 4          # neither skip belongs to frame D
    # (4) keep_optimizer_state (--smooth) keeps the step counter and the floor where they are
    eng.guard[0] = 5
    eng.reset_params(p0, keep_optimizer_state=True)
    assert eng._guard_floor == 4 and eng._state['step'][0] == 5
    look(4)
    assert eng._state['step'][0] == 4
    # the step counter never goes below zero
    eng._state['step'][0] = 0
    eng.guard[0] = 9
    look(5)
    assert eng._state['step'][0] == 0
