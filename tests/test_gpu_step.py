import os
"""GPU (MI355X): the whole optimisation step through the product library -- drop-in autograd API,
fused engine, loss-curve parity with the oracle, and size-independent properties at BASELINE's sizes."""
import warnings

import numpy as np
import pytest
import torch

from aphantasia_amd import _ffi, ops
from oracle import reference_path as R
from oracle import clip_vit_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


@pytest.fixture(scope='module')
def model():
    from aphantasia_amd import clip as aclip
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m, _ = aclip.load('ViT-B/32', seed=1, max_batch=16)
    return m


def oracle_run(model, h, w, target, params0, **kw):
    cfg, wts = model.visual.cfg, model.visual.weights
    return R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=params0, **kw)


def test_dropin_api_one_step_vs_oracle(model):
    """reference-style user code: fft_image / to_valid_rgb / slice_imgs / encode_image / sim_func / torch.optim.Adam"""
    from aphantasia_amd.image import fft_image, to_valid_rgb
    from aphantasia_amd.utils import slice_imgs, sim_func
    from aphantasia_amd import transforms
    h, w, S = 256, 320, 4
    seed_all(0)
    params, image_f, _ = fft_image([1, 3, h, w], 0.07, 1.5, None)
    rgb_f = to_valid_rgb(image_f, colors=1.8)
    opt = torch.optim.Adam(params, 0.05, betas=(.0, .999))
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    run = oracle_run(model, h, w, target, params[0].detach().cpu())
    seed_all(5)
    img = rgb_f()
    cuts = slice_imgs([img], S, 224, transforms.normalize(), 'uniform', 0.4)[0]
    enc = model.encode_image(cuts)
    loss = -1.0 * sim_func(target.to(DEV), enc, 'mix')
    opt.zero_grad()
    loss.backward()
    grad = params[0].grad.detach().cpu().clone()
    opt.step()
    seed_all(5)
    table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
    want = run.step(table)
    assert abs(float(loss) - want) < 5e-4
    ref = run.params.grad
    assert (grad - ref).abs().max().item() < 5e-2 * ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(grad.flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.999, cos
    # saved-frame path: image_f(contrast=1.1) under no_grad (clip_fft.py:299)
    with torch.no_grad():
        out = rgb_f(contrast=1.1).cpu()
        wantimg = run.image(1.1)
    assert (out - wantimg).abs().max().item() < 0.1 and (out - wantimg).pow(2).mean().sqrt().item() < 5e-3


def test_plain_image_f_and_pixel_image_grads():
    from aphantasia_amd.image import fft_image, pixel_image, to_valid_rgb
    h, w = 48, 64
    seed_all(1)
    params, image_f, _ = fft_image([1, 3, h, w], 0.07, 1.5, None)
    gw = torch.randn(1, 3, h, w)
    (image_f(contrast=1.2) * gw.to(DEV)).sum().backward()
    p = params[0].detach().cpu().requires_grad_(True)
    want = R.std_normalise(R.fft_image_raw(p, R.fft_scale(h, w, 1.5), h, w), 1.2)
    (want * gw).sum().backward()
    assert (params[0].grad.cpu() - p.grad).abs().max().item() < 1e-4 * p.grad.abs().max().item()
    prm, pix_f, _ = pixel_image([1, 3, h, w], sd=1.0)
    rgb_f = to_valid_rgb(pix_f, colors=2.0)
    (rgb_f(contrast=0.9) * gw.to(DEV)).sum().backward()
    q = prm[0].detach().cpu().requires_grad_(True)
    (R.synth_pixel(q, R.colcorr_t(2.0), 0.9) * gw).sum().backward()
    assert (prm[0].grad.cpu() - q.grad).abs().max().item() < 1e-4 * q.grad.abs().max().item()


def test_loss_curve_vs_oracle_free_running(model):
    """10 free-running Adam steps, 360x640, 8 cuts: per-step |dloss| <= 1e-3 (north_star tolerance)
    and final-image pixel RMS stated."""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    h, w, S, steps = 360, 640, 8, 10
    seed_all(0)
    p0 = R.fft_params_init([1, 3, h, w])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    eng = Engine(p0.to(DEV).contiguous(), h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.normalize())
    run = oracle_run(model, h, w, target, p0)
    seed_all(9)
    worst = 0.0
    for i in range(steps):
        table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
        got, want = float(eng.step(table)), run.step(table)
        worst = max(worst, abs(got - want))
        assert abs(got - want) < 1e-3, (i, got, want)
    with torch.no_grad():
        rms = (eng.synthesize(1.1).cpu() - run.image(1.1)[0]).pow(2).mean().sqrt().item()
    print('loss-curve max |d| %.2e, final pixel RMS %.4f' % (worst, rms))
    assert rms < 0.05


def test_engine_fast_transform_and_dualmodel_smoke(model):
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    h, w, S = 360, 640, 12
    seed_all(0)
    p0 = R.fft_params_init([1, 3, h, w]).to(DEV).contiguous()
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    eng = Engine(p0, h, w, model, S, [(target, -1.0), (target.flip(1), 0.5)], sim='mix', transform=transforms.transforms_fast)
    l0 = float(eng.step())
    for _ in range(8):
        l = float(eng.step())
    assert np.isfinite(l) and l < l0            # optimising: the loss goes down


def test_full_size_properties(model):
    """BASELINE size (1280x720, 190 cuts): determinism and adjoint identities, no oracle needed."""
    H, W, S = 720, 1280, 190
    seed_all(3)
    table, _ = __import__('aphantasia_amd.utils', fromlist=['x']).draw_crop_params(S, 224, H, W, 'uniform', 0.4)
    tb = torch.from_numpy(table).to(DEV)
    geom = ops.make_geom(H, W, S, 224, 32)
    img = torch.rand(3, H, W, device=DEV)
    y = ops.sample_fwd(geom, img, tb, out_mode=_ffi.APH_OUT_NCHW_NORM)
    g = torch.randn_like(y)
    gi = ops.sample_bwd(geom, g, tb, out_mode=_ffi.APH_OUT_NCHW_NORM)
    gi2 = ops.sample_bwd(geom, g, tb, out_mode=_ffi.APH_OUT_NCHW_NORM)
    assert torch.equal(gi, gi2)                                         # bitwise deterministic gather adjoint
    # <J x, g> == <x, J^T g> for the (affine) sampler: use differences to cancel the normalisation offset
    img2 = torch.rand(3, H, W, device=DEV)
    y2 = ops.sample_fwd(geom, img2, tb, out_mode=_ffi.APH_OUT_NCHW_NORM)
    lhs = ((y - y2).double() * g.double()).sum().item()
    rhs = ((img - img2).double() * gi.double()).sum().item()
    # (the inner product cancels heavily -- |lhs| is ~2e-4 of the sum of its terms' magnitudes -- so the bound is set by the norms)
    bound = 1e-6 * (y - y2).double().norm().item() * g.double().norm().item()
    assert abs(lhs - rhs) < bound, (lhs, rhs, bound)
    # synthesis adjoint: <d rgb, g> along a random parameter direction (finite difference, fp32)
    plan = ops.SynthPlan(3, H, W)
    seed_all(4)
    p = (0.01 * torch.randn(3, H, W // 2 + 1, 2)).to(DEV)
    scale = R.fft_scale(H, W, 1.5).to(DEV)
    cc = R.colcorr_t(1.8).flatten().tolist()
    raw, rgb = ops.synth_fft_fwd(plan, p, scale, None, 1.0, cc)
    gw = torch.randn_like(rgb)
    grad = ops.synth_fft_bwd(plan, gw, rgb, raw, scale, 1.0, cc)
    d = torch.randn_like(p)
    eps = 1e-3 * 0.01
    _, rp = ops.synth_fft_fwd(plan, p + eps * d, scale, None, 1.0, cc)
    _, rm = ops.synth_fft_fwd(plan, p - eps * d, scale, None, 1.0, cc)
    fd = (((rp - rm).double() / (2 * eps)) * gw.double()).sum().item()
    an = (grad.double() * d.double()).sum().item()
    assert abs(fd - an) < 2e-2 * abs(an), (fd, an)


def test_dwt_and_pixel_engines_vs_autograd_api(model):
    """dwt_image / pixel_image through the fused engine == the same step through the drop-in autograd API
    (orchestration check: both sides run the same kernels; parity with the oracle is in test_gpu_parity_configs.py)"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.image import dwt_image, pixel_image, to_valid_rgb
    from aphantasia_amd.utils import slice_imgs, sim_func
    from aphantasia_amd import transforms
    h, w, S = 256, 320, 4
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    for kind in ('dwt', 'pixel'):
        seed_all(0)
        if kind == 'dwt':
            params, image_f, _ = dwt_image([1, 3, h, w], 'db3', 0.3, 1.8, None)
        else:
            params, image_f, _ = pixel_image([1, 3, h, w], sd=1.0)
        rgb_f = to_valid_rgb(image_f, colors=1.8)
        seed_all(5)
        cuts = slice_imgs([rgb_f()], S, 224, transforms.normalize(), 'uniform', 0.4)[0]
        loss = -1.0 * sim_func(target.to(DEV), model.encode_image(cuts), 'mix')
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in params])
        if kind == 'dwt':
            eng = Engine(image_f.flat.detach().clone(), h, w, model, S, [(target, -1.0)], transform=transforms.normalize(),
                         param_kind='dwt', dwt=image_f.synth, rng='reference')
        else:
            eng = Engine(params[0].detach().clone(), h, w, model, S, [(target, -1.0)], transform=transforms.normalize(), param_kind='pixel', rng='reference')
        seed_all(5)
        l2 = float(eng.step())
        assert abs(l2 - float(loss)) < 1e-5, kind
        assert (eng.grad.reshape(-1) - grads).abs().max().item() < 1e-4 * grads.abs().max().item() + 1e-9, kind


def test_illustrip_rgb_step_priors_and_fixcontrast(model):
    """illustrip.py:425-440 inner step with `--gen RGB`: pixel_image(fixcontrast) + the brightness / contrast priors, fused
    engine (aph_rgb_priors) vs the same arithmetic written with torch ops on top of the drop-in autograd API"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.image import pixel_image, to_valid_rgb
    from aphantasia_amd.utils import slice_imgs, sim_func
    from aphantasia_amd import transforms
    h, w, S = 256, 320, 4
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    for fix in (False, True):
        seed_all(0)
        params, image_f, _ = pixel_image([1, 3, h, w], sd=1.0)
        rgb_f = to_valid_rgb(image_f, colors=1.8)
        seed_all(5)
        img_out = rgb_f(fixcontrast=fix)
        cuts = slice_imgs([img_out], S, 224, transforms.normalize(), 'uniform', 0.4)[0]
        loss = abs(img_out.mean((2, 3)) - 0.45).mean() + abs(img_out.std((2, 3)) - 0.17).mean()
        loss = loss - sim_func(target.to(DEV), model.encode_image(cuts), 'mix')
        loss.backward()
        eng = Engine(params[0].detach().clone(), h, w, model, S, [(target, -1.0)], transform=transforms.normalize(), param_kind='pixel',
                     rng='reference', rgb_priors=True, fixcontrast=fix)
        seed_all(5)
        l2 = float(eng.step())
        assert abs(l2 - float(loss)) < 2e-5, (fix, l2, float(loss))
        g = params[0].grad.reshape(-1)
        assert (eng.grad.reshape(-1) - g).abs().max().item() < 1e-4 * g.abs().max().item() + 1e-9, fix


def test_cli_resume_from_image_and_pt(tmp_path):
    """clip_fft.py --resume <image> (img2fft init, size from the image, align -> overscan) then --resume <.pt snapshot>"""
    import numpy as np
    from PIL import Image
    import clip_fft
    rng = np.random.default_rng(0)
    img = (rng.random((96, 128, 3)) * 255).astype(np.uint8)
    jpg = os.path.join(tmp_path, 'start.png')
    Image.fromarray(img).save(jpg)
    out = os.path.join(tmp_path, 'out')
    clip_fft.main(['-t', 'cat', '-nv', '--seed', '0', '--steps', '3', '--samples', '12', '--resume', jpg, '--out_dir', out, '--save_pt'])
    pts = [f for f in os.listdir(out) if f.endswith('.pt')]
    assert len(pts) == 1
    snap = torch.load(os.path.join(out, pts[0]))
    assert isinstance(snap, list) and tuple(snap[0].shape) == (1, 3, 96, 65, 2)          # clip_fft.py:315 list-of-tensors format
    frames = [f for f in os.listdir(os.path.join(out, os.path.splitext(pts[0])[0])) if f.endswith('.jpg')]
    assert len(frames) == 3
    clip_fft.main(['-t', 'cat', '-nv', '--seed', '0', '--steps', '2', '--samples', '12', '--size', '128-96', '--resume', os.path.join(out, pts[0]),
                   '--out_dir', os.path.join(tmp_path, 'out2'), '--no_save'])
    # wavelet parameters from the same image (img2dwt): the first synthesised frame reproduces the image's colours roughly
    clip_fft.main(['-t', 'cat', '-nv', '--seed', '0', '--steps', '2', '--samples', '12', '--dwt', '-w', 'db3', '--resume', jpg,
                   '--out_dir', os.path.join(tmp_path, 'out3'), '--no_save'])


def test_sharp_and_expand_terms_vs_autograd_api(model):
    """clip_fft.py:269-270 (--sharp) and :276-280 (--expand) in the fused engine vs the same terms written with torch ops
    on the drop-in autograd API, three Adam steps (orchestration check: same kernels on both sides; the oracle-based
    version is test_gpu_parity_configs.py::test_sharp_expand_terms_vs_oracle)"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.image import fft_image, to_valid_rgb
    from aphantasia_amd.utils import slice_imgs, sim_func
    from aphantasia_amd import transforms
    h, w, S, sharp, expand = 256, 320, 4, 0.6, 0.5
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))

    def derivat_naiv(img):                      # utils.py:265-268
        dx = torch.mean(torch.abs(img[:, :, :, 1:] - img[:, :, :, :-1]))
        dy = torch.mean(torch.abs(img[:, :, 1:, :] - img[:, :, :-1, :]))
        return 0.5 * (dx + dy)

    seed_all(0)
    params, image_f, _ = fft_image([1, 3, h, w], 0.07, 1.5, None)
    p0 = params[0].detach().clone()
    rgb_f = to_valid_rgb(image_f, colors=1.8)
    opt = torch.optim.Adam(params, 0.05, betas=(0.0, 0.999))
    want, prev = [], None
    for i in range(3):
        seed_all(10 + i)
        img_out = rgb_f()
        out_enc = model.encode_image(slice_imgs([img_out], S, 224, transforms.normalize(), 'uniform', 0.4)[0])
        loss = -1.0 * sim_func(target.to(DEV), out_enc, 'mix') - sharp * derivat_naiv(img_out)
        if i > 0:
            loss = loss + expand * sim_func(prev, out_enc, 'mix')
        prev = out_enc.detach().clone()
        opt.zero_grad(); loss.backward(); opt.step()
        want.append(float(loss))
    eng = Engine(p0.clone(), h, w, model, S, [(target, -1.0)], transform=transforms.normalize(), rng='reference', sharp=sharp, expand=expand)
    got = []
    for i in range(3):
        seed_all(10 + i)
        got.append(float(eng.step()))
        eng.set_prev_enc()
    assert np.abs(np.array(got) - np.array(want)).max() < 2e-4, (got, want)
    d = (eng.params.reshape(-1) - params[0].detach().reshape(-1)).abs()
    assert d.mean().item() < 2e-4, d.mean().item()


@pytest.mark.parametrize('tf', ['none', 'fast'])
def test_enforce_term_vs_autograd_api(model, tf):
    """clip_fft.py:271-275 (--enforce): second independently drawn set of cuts, pairwise similarity with gradient into both
    encodings -- fused engine (forward B, backward B, recompute A, backward A) vs torch autograd over the drop-in API
    (orchestration check; oracle-based: test_gpu_parity_configs.py::test_enforce_term_vs_oracle)"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.image import fft_image, to_valid_rgb
    from aphantasia_amd.utils import slice_imgs, sim_func
    from aphantasia_amd import transforms
    h, w, S, enforce = 256, 320, 4, 0.7
    trf = transforms.normalize() if tf == 'none' else transforms.transforms_fast
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    seed_all(0)
    params, image_f, _ = fft_image([1, 3, h, w], 0.07, 1.5, None)
    p0 = params[0].detach().clone()
    rgb_f = to_valid_rgb(image_f, colors=1.8)
    seed_all(21)
    out_enc = model.encode_image(slice_imgs([rgb_f()], S, 224, trf, 'uniform', 0.4)[0])
    loss = -1.0 * sim_func(target.to(DEV), out_enc, 'mix')
    out_enc2 = model.encode_image(slice_imgs([rgb_f()], S, 224, trf, 'uniform', 0.4)[0])
    loss = loss - enforce * sim_func(out_enc, out_enc2, 'mix')
    loss.backward()
    eng = Engine(p0.clone(), h, w, model, S, [(target, -1.0)], transform=trf, rng='reference', enforce=enforce)
    seed_all(21)
    got = float(eng.step())
    assert abs(got - float(loss)) < 1e-4, (got, float(loss))
    g = params[0].grad.reshape(-1)
    err = (eng.grad.reshape(-1) - g).abs().max().item()
    assert err < 2e-2 * g.abs().max().item(), (err, g.abs().max().item())


def test_step_is_bitwise_reproducible(model):
    """no atomics, fixed-order reductions, ordered split-K: two runs from the same seeds give the same bits (also a tripwire
    for LDS / barrier races in the pipelined GEMMs)"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))

    def run(graph):
        seed_all(0)
        h, w = 360, 640
        params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
        eng = Engine(params, h, w, model, 24, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph=graph)
        for _ in range(5):
            eng.step()
        torch.cuda.synchronize()
        return eng.params.clone(), eng.grad.clone(), eng.gpatch.clone(), float(eng.loss)
    a, b, c = run(False), run(False), run(True)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    for x, y in zip(a[:3], c[:3]):          # hipGraph replay == eager launches
        assert torch.equal(x, y)
    assert a[3] == b[3] == c[3]


def test_step_is_bitwise_reproducible_on_the_wave_specialised_gemm(model):
    """The same tripwire on the kernel that carries 92 of the 98 GEMM launches of the headline step: 100 cuts (M = 5000 token rows) with
    `aph_gemm_set_ws_min_tiles(1)`, so EVERY ViT GEMM of the step -- one and several tiles per workgroup, the operand ring running across
    output tiles -- goes through gemm_ws_kernel (vit_gemm_ws.h).  Its counted vmcnt waits / ring hand-offs are what the host interpreter
    cannot see: two eager runs and one hipGraph run of 5 steps must agree bit for bit."""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    L = _ffi.lib()
    model.visual.ensure_batch(100)
    prev = L.cdll.aph_gemm_set_ws_min_tiles(1)
    try:
        def run(graph):
            seed_all(0)
            h, w = 360, 640
            params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
            eng = Engine(params, h, w, model, 100, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, use_graph=graph)
            for _ in range(5):
                eng.step()
            torch.cuda.synchronize()
            return eng.params.clone(), eng.grad.clone(), eng.gpatch.clone(), float(eng.loss)
        a, b, c = run(False), run(False), run(True)
    finally:
        L.cdll.aph_gemm_set_ws_min_tiles(prev)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    for x, y in zip(a[:3], c[:3]):          # hipGraph replay == eager launches
        assert torch.equal(x, y)
    assert a[3] == b[3] == c[3] and np.isfinite(a[3])


def test_f16_patch_gradient_option_is_close(model):
    """Engine(grad_f16=True): the ViT input-gradient crosses to the sampler adjoint as loss-scaled f16 (aph_vit_backward_h +
    APH_GRAD_PATCH_F16); one step stays within f16 rounding of the default f32 hand-over"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    outs = []
    for f16 in (False, True):
        seed_all(0)
        h, w = 360, 640
        params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
        eng = Engine(params, h, w, model, 12, [(target, -1.0)], sim='mix', transform=transforms.transforms_fast, macro=0.4, grad_f16=f16, use_graph=False)
        loss = float(eng.step())
        outs.append((loss, eng.grad.clone()))
    assert outs[0][0] == outs[1][0]
    d = (outs[0][1] - outs[1][1]).abs().max().item()
    assert d < 2e-3 * outs[0][1].abs().max().item(), d


@pytest.mark.parametrize('gen', ['RGB', 'FFT'])
def test_illustrip_frame_loop_reparameterisation(model, gen):
    """illustrip.py:381-423: per frame, warp the current image (frame_transform), re-create the parameters from it and
    restart the optimiser.  Engine.reset_params does that in place (no re-allocation, hipGraphs stay valid) and must be
    indistinguishable from building a fresh engine on the warped parameters."""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    h, w, S = 256, 320, 6
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    kw = dict(sim='mix', transform=transforms.transforms_fast, macro=0.4, rng='reference')
    if gen == 'RGB':
        kw.update(param_kind='pixel', rgb_priors=True)
    seed_all(0)
    p0 = (torch.randn(1, 3, h, w) * 0.3 if gen == 'RGB' else 0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
    eng = Engine(p0.clone(), h, w, model, S, [(target, -1.0)], **kw)

    def warp(params):
        if gen == 'RGB':
            return transforms.frame_transform(params, (h, w), 2.0, (3, -1), 1.03, 1.0)
        img = torch.fft.irfftn(torch.view_as_complex(params), s=(h, w), norm='ortho')                  # illustrip.py:401-403
        img = transforms.frame_transform(img, (h, w), 2.0, (3, -1), 1.03, 1.0)
        return torch.view_as_real(torch.fft.rfftn(img, s=(h, w), dim=[2, 3], norm='ortho')).contiguous()   # :407-408

    for frame in range(3):
        for i in range(4):                       # enough steps for the hipGraph to be captured and replayed
            seed_all(100 * frame + i)
            l_last = float(eng.step())
        assert np.isfinite(l_last)
        new = warp(eng.params.detach())
        eng.reset_params(new)
        assert torch.equal(eng.params.reshape(-1), new.reshape(-1)) and eng.step_count == 0
        fresh = Engine(new.clone(), h, w, model, S, [(target, -1.0)], use_graph=False, **kw)
        seed_all(999); a = float(eng.step())
        seed_all(999); b = float(fresh.step())
        assert a == b, (frame, a, b)
        assert torch.equal(eng.params, fresh.params)
        eng.reset_params(new)                    # continue the loop from the warped frame


def test_loss_scale_backs_off_after_fp16_overflow(model):
    """an absurd loss scale overflows the fp16 backward: the guarded Adam skips those steps (parameters stay finite), the host
    halves the scale until the gradient is finite again, and the optimisation then proceeds"""
    from aphantasia_amd.engine import Engine
    from aphantasia_amd import transforms
    seed_all(0)
    h, w = 256, 320
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    params = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).to(DEV).contiguous()
    eng = Engine(params, h, w, model, 6, [(target, -1.0)], sim='mix', transform=transforms.normalize(), loss_scale=2.0 ** 30)
    eng.GUARD_EVERY = 2
    p0 = eng.params.clone()
    first = float(eng.step())
    assert torch.equal(eng.params, p0)                       # the very first step overflowed and was skipped
    for _ in range(80):
        last = float(eng.step())
    assert eng.loss_scale < 2.0 ** 30 and int(eng.guard[0]) > 0
    assert torch.isfinite(eng.params).all() and not torch.equal(eng.params, p0)
    assert last < first - 0.02, (first, last, eng.loss_scale)


@pytest.mark.parametrize('gen', ['RGB', 'FFT'])
def test_illustrip_cli_frames_with_and_without_depth(tmp_path, monkeypatch, gen):
    """illustrip.py end to end (synthetic CLIP weights): frames on disk, no skipped steps; `-d` runs the depth warp with a stand-in
    estimator (the CLI's own InferDepthAny needs a local Depth-Anything checkpoint and must refuse to start without one)"""
    import illustrip
    from aphantasia_amd import depthwarp
    from oracle import depth_ref
    out = str(tmp_path / 'o')
    base = ['-t', 'a cat', '-nv', '--seed', '0', '--steps', '3', '--samples', '12', '--size', '320-256', '--gen', gen, '--out_dir', out]
    illustrip.main(base)
    frames = [f for d, _, fs in os.walk(out) for f in fs if f.endswith('.jpg')]
    assert len(frames) == 3
    with pytest.raises(RuntimeError, match='Depth-Anything'):
        illustrip.main(base + ['-d', '0.3'])

    class ToyDepth:
        def __init__(self, *a, **k): pass
        def __call__(self, image): return depth_ref.toy_depth(image.cpu()).to(image.device)
    monkeypatch.setattr(depthwarp, 'InferDepthAny', ToyDepth)
    out2 = str(tmp_path / 'o2')
    illustrip.main(base[:-1] + [out2, '-d', '0.3', '--depth_dir', str(tmp_path / 'dm')])
    assert len([f for d, _, fs in os.walk(out2) for f in fs if f.endswith('.jpg')]) == 3
    assert len(os.listdir(str(tmp_path / 'dm'))) == 3              # one depth map per frame (depth.py:80-82)


def test_frame_writer_matches_checkout_arithmetic(tmp_path):
    """clip_fft.FrameWriter (aph_rgb_to_u8 on the step's stream + device->host copy on a side stream + JPEG threads) against the
    reference's own conversion utils.checkout (utils.py:94-100): np.clip(img ** gamma * 255, 0, 255).astype(np.uint8), HWC."""
    import clip_fft
    from PIL import Image
    h, w = 90, 130
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(3, h, w, generator=g) * 1.2 - 0.1).to(DEV).contiguous()            # some values outside [0, 1]
    wr = clip_fft.FrameWriter(h, w)
    for i, gamma in enumerate((1.0, 1.3)):
        x = img.clamp_min(0.0) if gamma != 1.0 else img                                   # (pow of a negative number is NaN upstream too)
        wr.put(x, os.path.join(tmp_path, '%d.png' % i), gamma)
        want = np.clip((x.cpu().numpy().astype(np.float32) ** np.float32(gamma)) * 255, 0, 255).astype(np.uint8).transpose(1, 2, 0)
        torch.cuda.synchronize()
        got = wr.dev[(wr.n - 1) % wr.RING].cpu().numpy()
        assert np.abs(got.astype(int) - want.astype(int)).max() <= (0 if gamma == 1.0 else 1), gamma      # powf vs numpy pow: last-ulp at a truncation boundary
    wr.drain()
    wr.close()
    for i in range(2):
        back = np.asarray(Image.open(os.path.join(tmp_path, '%d.png' % i)))                # PNG: lossless, what the pinned buffer held
        assert back.shape == (h, w, 3)
    assert np.array_equal(np.asarray(Image.open(os.path.join(tmp_path, '0.png'))), np.clip(img.cpu().numpy() * 255, 0, 255).astype(np.uint8).transpose(1, 2, 0))


def test_frame_writer_ring_reuse_keeps_every_frame(tmp_path):
    """[r5] More frames than the writer has ring slots, put back to back without draining: a slot's device image and pinned buffer are
    rewritten only after the copy and the encoder thread of the frame that last used them are done (per-slot event + semaphore), so every
    file on disk holds ITS frame (lossless PNG, frame i is the constant i)."""
    import clip_fft
    from PIL import Image
    h, w = 64, 96
    wr = clip_fft.FrameWriter(h, w)
    n = 3 * wr.RING + 5
    for i in range(n):
        img = torch.full((3, h, w), (i + 0.5) / 255.0, device=DEV)
        wr.put(img, os.path.join(tmp_path, '%03d.png' % i), 1.0)
    wr.drain()
    wr.close()
    for i in range(n):
        back = np.asarray(Image.open(os.path.join(tmp_path, '%03d.png' % i)))
        assert back.shape == (h, w, 3) and int(back.min()) == i and int(back.max()) == i, i
