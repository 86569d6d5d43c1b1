"""CPU: the product's HIP kernel sources executed under the host SIMT interpreter (tests/emu) through
the C ABI, checked against the oracle / reference-generated goldens.  This is a logic check of the
kernels (indexing, LDS staging, barriers, reductions) -- numerics on real gfx950 hardware are
covered by the `-m gpu` tests, which call the same C ABI in libaphantasia_hip.so."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from aphantasia_amd import _ffi, ops
from oracle import reference_path as R
from oracle import clip_vit_ref, augment_ref

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))


@pytest.fixture(scope='module')
def emu():
    import build_emu
    return _ffi.Library(build_emu.build())


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


@pytest.mark.parametrize('name', ['synth_48x80.npz', 'synth_45x63.npz'])
def test_synth_fft_fwd_bwd(emu, golden, name):
    g = golden(name)
    h, w = int(g['h']), int(g['w'])
    params = torch.from_numpy(g['params']).contiguous()
    scale = R.fft_scale(h, w, float(g['decay'])).contiguous()
    cc = R.colcorr_t(float(g['colors'])).flatten().tolist()
    plan = ops.SynthPlan(3, h, w, lib=emu)
    contrast = float(g['contrast'])
    raw, rgb = ops.synth_fft_fwd(plan, params, scale, None, contrast, cc, True, lib=emu)
    std = raw.std()
    assert np.allclose((raw / std).numpy(), g['raw'][0], atol=2e-5)
    assert np.allclose(rgb.numpy(), g['rgb'][0], atol=2e-6)
    grad = ops.synth_fft_bwd(plan, torch.from_numpy(g['gw'][0]).contiguous(), rgb, raw, scale, contrast, cc, True, lib=emu)
    ref = g['grad'][0]
    assert np.abs(grad.numpy() - ref).max() < 2e-5 * np.abs(ref).max()


def test_synth_shift_and_spatial(emu):
    h, w = 24, 40
    seed_all(1)
    params = R.fft_params_init([1, 3, h, w])
    scale = R.fft_scale(h, w, 1.5)
    cc_t = R.colcorr_t(1.8)
    shift = 0.02 * torch.rand(1, 1, h, w // 2 + 1, 1)
    want = R.synth_fft(params, scale, h, w, cc_t, 1.0, shift)[0]
    plan = ops.SynthPlan(3, h, w, lib=emu)
    _, rgb = ops.synth_fft_fwd(plan, params.contiguous(), scale, shift.reshape(h, w // 2 + 1).contiguous(), 1.0,
                               cc_t.flatten().tolist(), True, lib=emu)
    assert (rgb - want).abs().max().item() < 3e-6
    # pixel_image parameteriser (image.py:114-118), both contrast modes
    for fix in (False, True):
        img = torch.randn(1, 3, h, w).requires_grad_(True)
        want = R.synth_pixel(img, cc_t, 1.1, fix)
        gw = torch.randn(1, 3, h, w)
        (want * gw).sum().backward()
        rgb = ops.synth_spatial_fwd(plan, img.detach()[0].contiguous(), 1.1, 3.3 if fix else 0.0, cc_t.flatten().tolist(), True, lib=emu)
        assert (rgb - want.detach()[0]).abs().max().item() < 2e-6
        d = ops.synth_spatial_bwd(plan, gw[0].contiguous(), rgb, img.detach()[0].contiguous(), 1.1, 3.3 if fix else 0.0,
                                  cc_t.flatten().tolist(), True, lib=emu)
        assert (d - img.grad[0]).abs().max().item() < 2e-5 * img.grad.abs().max().item() + 1e-7


@pytest.mark.parametrize('align', ['uniform', 'overscan', 'overmax'])
def test_sampler_vs_reference_golden(emu, golden, align):
    g = golden('slice_48x80.npz')
    img = torch.from_numpy(g['img'])
    seed_all(7)
    table = R.draw_crop_table(6, 16, 48, 80, align, 0.4)
    geom = ops.make_geom(48, 80, 6, 16, patch=8, align=align)
    out = ops.sample_fwd(geom, img[0].contiguous(), torch.from_numpy(table), lib=emu)
    assert np.abs(out.numpy() - g['cuts_' + align]).max() < 5e-6
    # patch-major f16 layout == patchify(NCHW)
    pm = ops.sample_fwd(geom, img[0].contiguous(), torch.from_numpy(table), out_mode=_ffi.APH_OUT_PATCH_F16, lib=emu)
    pm2 = ops.patchify(out, 8, lib=emu)
    assert torch.equal(pm, pm2)
    want = out.reshape(6, 3, 2, 8, 2, 8).permute(0, 2, 4, 1, 3, 5).reshape(6 * 4, 3 * 64).half()
    assert torch.equal(pm2, want)


@pytest.mark.parametrize('align,mode', [('uniform', _ffi.APH_OUT_NCHW_NORM), ('overscan', _ffi.APH_OUT_NCHW_RAW),
                                        ('uniform', _ffi.APH_OUT_PATCH_F16)])
def test_sampler_adjoint(emu, align, mode):
    seed_all(3)
    H, W, S, size = 40, 56, 5, 16
    img = torch.rand(1, 3, H, W).requires_grad_(True)
    table = R.draw_crop_table(S, size, H, W, align, 0.4)
    table[0] = (size - 3, 1, 2)          # one up-sampling cut (csize < size)
    cuts = R.slice_imgs(img, table, size, align, transform=None if mode == _ffi.APH_OUT_NCHW_RAW else R.normalize)
    gout = torch.randn_like(cuts)
    (cuts * gout).sum().backward()
    geom = ops.make_geom(H, W, S, size, patch=8, align=align)
    gin = gout
    if mode == _ffi.APH_OUT_PATCH_F16:
        gin = gout.reshape(S, 3, 2, 8, 2, 8).permute(0, 2, 4, 1, 3, 5).reshape(S * 4, 3 * 64)
    got = ops.sample_bwd(geom, gin.contiguous(), torch.from_numpy(table), out_mode=mode, gscale=2.0, lib=emu)
    assert (got - 2.0 * img.grad[0]).abs().max().item() < 1e-4 * img.grad.abs().max().item()


def test_sampler_augment_fwd_bwd(emu):
    seed_all(5)
    H, W, S, size = 40, 48, 6, 16
    img = torch.rand(1, 3, H, W).requires_grad_(True)
    prms = []
    table = R.draw_crop_table(S, size, H, W, 'uniform', 0.4)
    for s in range(S):
        sp, ep = augment_ref.perspective_get_params(size, size, 0.33)
        prms.append(dict(persp=augment_ref.perspective_coeffs(sp, ep) if s % 2 == 0 else None,
                         erase=(2, 3, 5, 7) if s in (1, 2) else None, angle=[-30.0, 0.0, 17.0, 29.0, -5.0, 0.0][s]))
    cuts = R.slice_imgs(img, table, size, 'uniform', per_cut=lambda c, cut: augment_ref.apply_fast(cut, prms[c], R.normalize))
    gout = torch.randn_like(cuts)
    (cuts * gout).sum().backward()
    from aphantasia_amd.transforms import pack_aug
    aug = pack_aug(prms)
    geom = ops.make_geom(H, W, S, size, patch=8)
    out = ops.sample_fwd(geom, img.detach()[0].contiguous(), torch.from_numpy(table), aug=aug, lib=emu)
    assert (out - cuts.detach()).abs().max().item() < 2e-4
    got = ops.sample_bwd(geom, gout.contiguous(), torch.from_numpy(table), aug=aug, lib=emu)
    assert (got - img.grad[0]).abs().max().item() < 2e-4 * img.grad.abs().max().item()


def test_sim_loss_vs_reference_golden(emu, golden):
    g = golden('sim.npz')
    v1 = torch.from_numpy(g['v1'])
    v2 = torch.from_numpy(g['v2']).contiguous()
    for t in [None, 'mix', 'ang', 'dot']:
        loss, genc = ops.sim_loss(v2, v1.contiguous(), [1.0], t, lib=emu)
        assert abs(loss.item() - float(g['val_%s' % t])) < 2e-6 * max(1.0, abs(float(g['val_%s' % t]))), t
        assert np.allclose(genc.numpy(), g['grad_%s' % t], rtol=2e-4, atol=2e-7), t
    # two weighted targets, sign -1, loss scale
    tg = torch.randn(2, 64)
    x = v2.clone().requires_grad_(True)
    want = -1.0 * R.sim_func(tg[0:1], x, 'mix') + 0.5 * R.sim_func(tg[1:2], x, 'mix')
    want.backward()
    loss, genc = ops.sim_loss(v2, tg, [-1.0, 0.5], 'mix', gscale=8.0, lib=emu)
    assert abs(loss.item() - want.item()) < 1e-6
    assert np.allclose(genc.numpy() / 8.0, x.grad.numpy(), rtol=2e-4, atol=2e-7)


def test_adam_variants(emu):
    gen = torch.Generator().manual_seed(0)
    for name, kw, dec, ams in [('adam_custom', dict(beta1=0.0), False, False), ('adam', dict(beta1=0.9), False, False),
                               ('adamw', dict(beta1=0.9, weight_decay=0.01), True, False),
                               ('adamw_custom', dict(beta1=0.0, weight_decay=0.01), True, True)]:
        p = torch.randn(5000, generator=gen)
        q = p.clone().requires_grad_(True)
        opt = R.make_optimizer([q], name, 0.05)
        m = torch.zeros_like(p) if kw['beta1'] else None
        v, vm = torch.zeros_like(p), (torch.zeros_like(p) if ams else None)
        for step in range(1, 4):
            grad = torch.randn(5000, generator=gen) * 0.01
            q.grad = grad.clone()
            opt.step()
            hyper = torch.tensor(ops.adam_hyper(step, 0.05, **kw), dtype=torch.float32)
            ops.adam_step(p, grad, m, v, vm, hyper, dec, lib=emu)
            assert (p - q.detach()).abs().max().item() < 2e-6, name


def test_gemm_core(emu):
    gen = torch.Generator().manual_seed(0)
    for (M, N, K) in [(100, 128, 64), (130, 256, 192)]:
        A = torch.randn(M, K, generator=gen).half()
        Bt = torch.randn(N, K, generator=gen).half()       # asymmetric operands (catches transposes)
        C = ops.gemm_f16(A, Bt, lib=emu)
        want = A.float() @ Bt.float().T
        assert (C - want).abs().max().item() < 1e-3


TINY = dict(input_resolution=32, patch_size=16, width=256, layers=2, heads=4, output_dim=128)


def test_vit_forward_backward(emu):
    from aphantasia_amd.weights import synthetic_visual_weights
    w = synthetic_visual_weights(TINY, 3)
    S = 3
    x = torch.randn(S, 3, 32, 32, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
    want = clip_vit_ref.encode_image(w, x, TINY)
    genc = torch.randn(S, 128, generator=torch.Generator().manual_seed(2)) * 0.01
    (want * genc).sum().backward()
    vit = ops.VitHandle(TINY, w, max_batch=4, lib=emu)
    patches = ops.patchify(x.detach().contiguous(), 16, lib=emu)
    enc = vit.forward(patches, S)
    assert (enc - want.detach()).abs().max().item() < 3e-3 * want.abs().max().item()
    LS = 1024.0
    gp = vit.backward((genc * LS).contiguous(), S, out_scale=1.0 / LS)
    gx = ops.unpatchify(gp, S, 32, 16, lib=emu)
    err = (gx - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert err < 2e-2, err
