"""Kernel checks shared by the CPU (tests/emu interpreter) and GPU (libaphantasia_hip.so) test files.
Each check drives the C ABI through aphantasia_amd.ops on tensors living on `dev` and compares with the
oracle (oracle/) or the reference-generated goldens.  lib=None selects the product library."""
import os

import numpy as np
import torch

from aphantasia_amd import _ffi, ops
from aphantasia_amd.transforms import pack_aug
from aphantasia_amd.weights import synthetic_visual_weights
from oracle import reference_path as R
from oracle import clip_vit_ref, augment_ref


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def to_patch_major(x, p):
    """[S,3,R,R] -> [S*g*g, 3*p*p] (the APH_OUT_PATCH_F16 element order: patch (gy, gx), then pixel-major inside the patch with the
    channel fastest -- k = (iy * p + ix) * 3 + c)"""
    S, _, Rr, _ = x.shape
    g = Rr // p
    return x.reshape(S, 3, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(S * g * g, 3 * p * p)


def check_synth_golden(lib, dev, g):
    h, w = int(g['h']), int(g['w'])
    params = torch.from_numpy(g['params']).to(dev).contiguous()
    scale = R.fft_scale(h, w, float(g['decay'])).to(dev).contiguous()
    cc = R.colcorr_t(float(g['colors'])).flatten().tolist()
    plan = ops.SynthPlan(3, h, w, lib=lib)
    contrast = float(g['contrast'])
    raw, rgb = ops.synth_fft_fwd(plan, params, scale, None, contrast, cc, True, lib=lib)
    raw_c, rgb_c = raw.cpu(), rgb.cpu()
    assert np.allclose((raw_c / raw_c.std()).numpy(), g['raw'][0], atol=2e-5)
    assert np.allclose(rgb_c.numpy(), g['rgb'][0], atol=2e-6)
    grad = ops.synth_fft_bwd(plan, torch.from_numpy(g['gw'][0]).to(dev).contiguous(), rgb, raw, scale, contrast, cc, True, lib=lib)
    ref = g['grad'][0]
    assert np.abs(grad.cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()


def check_synth_vs_oracle(lib, dev, h, w, contrast=1.0, with_shift=False):
    """forward + adjoint at an arbitrary size against the torch-CPU oracle (autograd)."""
    seed_all(1)
    params = R.fft_params_init([1, 3, h, w]).requires_grad_(True)
    scale = R.fft_scale(h, w, 1.5)
    cc_t = R.colcorr_t(1.8)
    shift = 0.02 * torch.rand(1, 1, h, w // 2 + 1, 1) if with_shift else None
    want = R.synth_fft(params, scale, h, w, cc_t, contrast, shift)
    gw = torch.randn(1, 3, h, w)
    (want * gw).sum().backward()
    plan = ops.SynthPlan(3, h, w, lib=lib)
    sh = shift.reshape(h, w // 2 + 1).to(dev).contiguous() if with_shift else None
    raw, rgb = ops.synth_fft_fwd(plan, params.detach().to(dev).contiguous(), scale.to(dev), sh, contrast,
                                 cc_t.flatten().tolist(), True, lib=lib)
    assert (rgb.cpu() - want.detach()[0]).abs().max().item() < 4e-6
    grad = ops.synth_fft_bwd(plan, gw[0].to(dev).contiguous(), rgb, raw, scale.to(dev), contrast, cc_t.flatten().tolist(), True, lib=lib)
    ref = params.grad[0]
    assert (grad.cpu() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()


def check_fft_pair(lib, dev, h, w):
    """aph_irfft2 / aph_rfft2 == torch.fft.irfftn / rfftn with norm='ortho' (illustrip.py:401-409), and the round trip"""
    seed_all(6)
    plan = ops.SynthPlan(3, h, w, lib=lib)
    spec = torch.randn(1, 3, h, w // 2 + 1, 2)
    want = torch.fft.irfftn(torch.view_as_complex(spec), s=(h, w), norm='ortho')
    got = ops.irfft2(plan, spec.to(dev).contiguous(), lib=lib)
    assert (got.cpu() - want[0]).abs().max().item() < 2e-5 * want.abs().max().item()
    img = torch.randn(1, 3, h, w)
    wants = torch.view_as_real(torch.fft.rfftn(img, s=(h, w), dim=[2, 3], norm='ortho'))
    gots = ops.rfft2(plan, img[0].to(dev).contiguous(), lib=lib)
    assert (gots.cpu() - wants[0]).abs().max().item() < 2e-5 * wants.abs().max().item()
    back = ops.irfft2(plan, gots, lib=lib)
    assert (back.cpu() - img[0]).abs().max().item() < 2e-5 * img.abs().max().item()


def check_synth_spatial(lib, dev, h=24, w=40):
    seed_all(2)
    cc_t = R.colcorr_t(1.8)
    plan = ops.SynthPlan(3, h, w, lib=lib)
    for fix in (False, True):
        img = torch.randn(1, 3, h, w).requires_grad_(True)
        want = R.synth_pixel(img, cc_t, 1.1, fix)
        gw = torch.randn(1, 3, h, w)
        (want * gw).sum().backward()
        raw = img.detach()[0].to(dev).contiguous()
        rgb = ops.synth_spatial_fwd(plan, raw, 1.1, 3.3 if fix else 0.0, cc_t.flatten().tolist(), True, lib=lib)
        assert (rgb.cpu() - want.detach()[0]).abs().max().item() < 2e-6
        d = ops.synth_spatial_bwd(plan, gw[0].to(dev).contiguous(), rgb, raw, 1.1, 3.3 if fix else 0.0,
                                  cc_t.flatten().tolist(), True, lib=lib)
        assert (d.cpu() - img.grad[0]).abs().max().item() < 2e-5 * img.grad.abs().max().item() + 1e-7


def check_dwt(lib, dev, wave, h, w, sharp=0.3, colors=1.5, contrast=1.1):
    """dwt_image + to_valid_rgb: inverse DWT levels, std, colour, sigmoid and the full adjoint vs the oracle"""
    from aphantasia_amd.dwt import DWTSynth
    from oracle import dwt_ref
    seed_all(8)
    Ys = [y.requires_grad_(True) for y in dwt_ref.init_params([1, 3, h, w], wave)]
    cc_t = R.colcorr_t(colors)
    want = dwt_ref.synth_dwt(Ys, wave, cc_t, sharp, contrast)
    gw = torch.randn(want.shape, generator=torch.Generator().manual_seed(3))
    (want * gw).sum().backward()
    syn = DWTSynth(h, w, wave, sharp, dev, lib=lib)
    assert [tuple(y.shape) for y in Ys] == list(syn.shapes)
    flat = torch.cat([y.detach().reshape(-1) for y in Ys]).to(dev).contiguous()
    raw = syn.forward(flat)
    assert tuple(raw.shape[1:]) == tuple(want.shape[2:])
    plan = ops.SynthPlan(3, syn.H, syn.W, lib=lib)
    rgb = ops.synth_spatial_fwd(plan, raw, contrast, 0.0, cc_t.flatten().tolist(), True, lib=lib)
    assert (rgb.cpu() - want.detach()[0]).abs().max().item() < 5e-6
    d_raw = ops.synth_spatial_bwd(plan, gw[0].to(dev).contiguous(), rgb, raw, contrast, 0.0, cc_t.flatten().tolist(), True, lib=lib)
    grad = torch.empty_like(flat)
    syn.backward(d_raw, grad)
    ref = torch.cat([y.grad.reshape(-1) for y in Ys])
    assert (grad.cpu() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    # one launch per level (aph_idwt_level_*) computes the same sums in the same order as the all-levels calls, whose coarse
    # tail runs in one launch (bit-identical under the interpreter; on the GPU the two instantiations may contract their
    # multiply-adds differently, so a rounding-level tolerance there)
    raw1 = raw.clone()
    raw2 = syn.forward_per_level(flat)
    grad1 = torch.full_like(flat, float('nan'))
    syn.backward_per_level(d_raw, grad1)
    if dev == 'cpu':
        assert torch.equal(raw2, raw1) and torch.equal(grad1, grad)
    else:
        assert (raw2 - raw1).abs().max().item() <= 1e-6 * raw1.abs().max().item()
        assert (grad1 - grad).abs().max().item() <= 1e-6 * grad.abs().max().item()


def check_sampler_golden(lib, dev, g, align):
    img = torch.from_numpy(g['img'])
    seed_all(7)
    table = R.draw_crop_table(6, 16, 48, 80, align, 0.4)
    geom = ops.make_geom(48, 80, 6, 16, patch=8, align=align)
    tb = torch.from_numpy(table).to(dev)
    out = ops.sample_fwd(geom, img[0].to(dev).contiguous(), tb, lib=lib)
    err = np.abs(out.cpu().numpy() - g['cuts_' + align]).max()
    assert err < 3e-5, err        # normalised values reach +-2; fp32 contraction order differs from ATen's
    pm = ops.sample_fwd(geom, img[0].to(dev).contiguous(), tb, out_mode=_ffi.APH_OUT_PATCH_F16, lib=lib)
    pm2 = ops.patchify(out, 8, lib=lib)
    assert torch.equal(pm, pm2)
    assert torch.equal(pm2.cpu(), to_patch_major(out.cpu(), 8).half())


def check_sampler_adjoint(lib, dev, align, mode, H=40, W=56, S=5, size=16, patch=8):
    seed_all(3)
    img = torch.rand(1, 3, H, W).requires_grad_(True)
    table = R.draw_crop_table(S, size, H, W, align, 0.4)
    if min(H, W) > size + 3:
        table[0] = (size - 3, 1, 2)          # one up-sampling cut (csize < size)
    cuts = R.slice_imgs(img, table, size, align, transform=None if mode == _ffi.APH_OUT_NCHW_RAW else R.normalize)
    gout = torch.randn_like(cuts)
    (cuts * gout).sum().backward()
    geom = ops.make_geom(H, W, S, size, patch=patch, align=align)
    gin = to_patch_major(gout, patch) if mode == _ffi.APH_OUT_PATCH_F16 else gout
    got = ops.sample_bwd(geom, gin.to(dev).contiguous(), torch.from_numpy(table).to(dev), out_mode=mode, gscale=2.0, lib=lib)
    assert (got.cpu() - 2.0 * img.grad[0]).abs().max().item() < 1e-4 * img.grad.abs().max().item()


def check_sampler_augment(lib, dev, H=40, W=48, S=6, size=16, patch=8):
    seed_all(5)
    img = torch.rand(1, 3, H, W).requires_grad_(True)
    prms = []
    table = R.draw_crop_table(S, size, H, W, 'uniform', 0.4)
    angles = [-30.0, 0.0, 17.0, 29.0, -5.0, 0.0]
    for s in range(S):
        sp, ep = augment_ref.perspective_get_params(size, size, 0.33)
        prms.append(dict(persp=augment_ref.perspective_coeffs(sp, ep) if s % 2 == 0 else None,
                         erase=(2, 3, size // 3, size // 2) if s % 6 in (1, 2) else None, angle=angles[s % 6]))
    cuts = R.slice_imgs(img, table, size, 'uniform', per_cut=lambda c, cut: augment_ref.apply_fast(cut, prms[c], R.normalize))
    gout = torch.randn_like(cuts)
    (cuts * gout).sum().backward()
    aug = pack_aug(prms).to(dev)
    geom = ops.make_geom(H, W, S, size, patch=patch)
    tb = torch.from_numpy(table).to(dev)
    out = ops.sample_fwd(geom, img.detach()[0].to(dev).contiguous(), tb, aug=aug, lib=lib)
    assert (out.cpu() - cuts.detach()).abs().max().item() < 3e-4
    # the patch-major f16 emit (LDS-staged 16-byte stores for full tiles) holds the same values
    pm = ops.sample_fwd(geom, img.detach()[0].to(dev).contiguous(), tb, aug=aug, out_mode=_ffi.APH_OUT_PATCH_F16, lib=lib)
    assert torch.equal(pm.cpu(), to_patch_major(out.cpu(), patch).half())
    # the split-precision rows [hi | lo] (APH_OUT_PATCH_F16_HILO): hi = the plain f16 emit, hi + lo = the f32 value to ~2^-22; with and without -tf fast
    for a_ in (aug, None):
        hl = ops.sample_fwd(geom, img.detach()[0].to(dev).contiguous(), tb, aug=a_, out_mode=_ffi.APH_OUT_PATCH_F16_HILO, lib=lib).cpu()
        f32 = to_patch_major(ops.sample_fwd(geom, img.detach()[0].to(dev).contiguous(), tb, aug=a_, lib=lib).cpu(), patch)
        kp = 3 * patch * patch
        assert hl.shape == (f32.shape[0], 2 * kp) and torch.equal(hl[:, :kp], f32.half())
        assert (hl[:, :kp].float() + hl[:, kp:].float() - f32).abs().max().item() < 2e-6 * max(f32.abs().max().item(), 1.0)
    got = ops.sample_bwd(geom, gout.to(dev).contiguous(), tb, aug=aug, lib=lib)
    assert (got.cpu() - img.grad[0]).abs().max().item() < 3e-4 * img.grad.abs().max().item()


TV_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_fast_224.npz')


def tv_fixture_or_skip():
    """the torchvision fixture (oracle/make_tv_fixture.py: needs torchvision, which neither the build image nor the GPU boxes have)"""
    import pytest
    if not os.path.isfile(TV_FIXTURE):
        pytest.skip('PARITY UNPINNED for transforms_fast / frame_transform: tests/golden/tf_fast_224.npz does not exist -- run '
                    '`python oracle/make_tv_fixture.py` on a machine that has torchvision and /root/reference, commit the file, and this '
                    'test pins SURVEY rows a-8 / f-1 against torchvision itself')
    return np.load(TV_FIXTURE)


def check_oracle_vs_tv_fixture(fx):
    """oracle/augment_ref.py (the restated torchvision maths) against outputs of torchvision itself: the seeded transforms_fast
    stream (op arithmetic + draw order), explicit perspectives, rotations and full affines"""
    from oracle import make_tv_fixture as F
    cuts, frame = F.inputs()
    seed_all(F.STREAM_SEED)
    got = torch.cat([augment_ref.apply_fast(cuts[c:c + 1], augment_ref.draw_fast_params(F.SIZE), R.normalize) for c in range(F.STREAM_CUTS)], 0)
    assert (got - torch.from_numpy(fx['stream_out'])).abs().max().item() < 2e-5
    for i, (sp, ep) in enumerate(F.PERSP_CASES):
        got = augment_ref.perspective(cuts[i:i + 1], augment_ref.perspective_coeffs(sp, ep))
        assert (got - torch.from_numpy(fx['persp_out'][i:i + 1])).abs().max().item() < 2e-5, i
    for i, (a, t, sc, sh) in enumerate(F.AFFINE_CASES):
        c = i % F.STREAM_CUTS
        assert (augment_ref.rotate(cuts[c:c + 1], a) - torch.from_numpy(fx['rotate_out'][i:i + 1])).abs().max().item() < 2e-5, i
        assert (augment_ref.affine(frame, a, t, sc, sh) - torch.from_numpy(fx['affine_out'][i:i + 1])).abs().max().item() < 2e-5, i


def check_kernels_vs_tv_fixture(lib, dev, fx):
    """the HIP sampler's augment stages and aph_frame_affine against torchvision's outputs: every cut is fed as a 224 x 224 image with
    an identity crop (bicubic taps at integer positions = exact copy), so what is compared is perspective / erase / rotate alone"""
    from oracle import make_tv_fixture as F
    from aphantasia_amd import transforms as T
    cuts, frame = F.inputs()
    size = F.SIZE
    geom = ops.make_geom(size, size, 1, size, patch=32)
    table = torch.tensor([[size, 0, 0]], dtype=torch.int32, device=dev)
    seed_all(F.STREAM_SEED)
    for c in range(F.STREAM_CUTS):
        prm = augment_ref.draw_fast_params(size)
        out = ops.sample_fwd(geom, cuts[c].to(dev).contiguous(), table, aug=pack_aug([prm]).to(dev), lib=lib).cpu()
        assert (out - torch.from_numpy(fx['stream_out'][c:c + 1])).abs().max().item() < 2e-4, (c, prm)
    for i, (a, t, sc, sh) in enumerate(F.AFFINE_CASES):
        got = T.frame_transform(frame.to(dev), F.FRAME_HW, a, t, sc, sh, lib=lib).cpu()
        assert (got - torch.from_numpy(fx['affine_out'][i:i + 1])).abs().max().item() < 2e-4, i


def check_kernels_vs_pil(lib, dev, size=64, patch=32):
    """the HIP sampler's perspective / rotation stages and aph_frame_affine against Pillow's float `Image.transform` (the implementation behind
    torchvision's PIL backend; interior pixels -- the border band blends differently by construction).  Every cut is a size x size image with an
    identity crop, so only the warp is compared."""
    from PIL import Image
    from aphantasia_amd import transforms as T
    g = torch.Generator().manual_seed(3)
    cut = torch.rand(3, size, size, generator=g)
    geom = ops.make_geom(size, size, 1, size, patch=patch)
    table = torch.tensor([[size, 0, 0]], dtype=torch.int32, device=dev)
    def covered(warp_ones):
        """pixels whose whole bilinear footprint lies inside the source (warped all-ones image == 1), eroded by one pixel: where the
        two implementations must agree; elsewhere Pillow fills and the tensor path blends"""
        m = (warp_ones[0, 0] > 1 - 1e-6).float()[None, None]
        m = -torch.nn.functional.max_pool2d(-m, 3, 1, 1)
        return m[0, 0].bool().numpy()
    ones = torch.ones(1, 1, size, size)
    start = [[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]]
    ends = ([[4, 6], [57, 2], [60, 58], [3, 52]], [[7, 1], [62, 5], [55, 61], [1, 50]])
    pil_of = lambda ch, mode, coef: np.asarray(Image.fromarray(cut[ch].numpy().astype(np.float32), mode='F').transform((size, size), mode, coef, Image.BILINEAR))
    for end in ends:
        co = augment_ref.perspective_coeffs(start, end)
        out = ops.sample_fwd(geom, cut.to(dev).contiguous(), table, aug=pack_aug([dict(persp=co, erase=None, angle=None)]).to(dev), out_mode=_ffi.APH_OUT_NCHW_RAW,
                             lib=lib).cpu()[0]
        want = np.stack([pil_of(ch, Image.PERSPECTIVE, co) for ch in range(3)])
        ok = covered(augment_ref.perspective(ones, co))
        assert ok.mean() > 0.5 and np.abs(out.numpy() - want)[:, ok].max() < 2e-4, end
    for ang in (30.0, -17.0, 5.0):
        out = ops.sample_fwd(geom, cut.to(dev).contiguous(), table, aug=pack_aug([dict(persp=None, erase=None, angle=ang)]).to(dev), out_mode=_ffi.APH_OUT_NCHW_RAW,
                             lib=lib).cpu()[0]
        a, b, c0, d, e, f0 = augment_ref.inverse_affine_matrix(ang, [0, 0], 1.0, 0.0)
        cx = cy = size * 0.5
        mat = [a, b, c0 + cx - a * cx - b * cy, d, e, f0 + cy - d * cx - e * cy]
        want = np.stack([pil_of(ch, Image.AFFINE, mat) for ch in range(3)])
        ok = covered(augment_ref.rotate(ones, ang))
        assert ok.mean() > 0.5 and np.abs(out.numpy() - want)[:, ok].max() < 2e-4, ang
    h, w = 48, 72
    frame = torch.rand(1, 3, h, w, generator=g)
    for (ang, t, sc, sh) in ((0.8, [0.0, 10.0], 1.012, 0.4), (-2.5, [7.0, -3.0], 0.97, -1.2)):
        got = T.frame_transform(frame.to(dev), (h, w), ang, t, sc, sh, lib=lib).cpu()[0]
        a, b, c0, d, e, f0 = augment_ref.inverse_affine_matrix(ang, t, sc, sh)
        cx, cy = w * 0.5, h * 0.5
        mat = [a, b, c0 + cx - a * cx - b * cy, d, e, f0 + cy - d * cx - e * cy]
        want = np.stack([np.asarray(Image.fromarray(frame[0, ch].numpy().astype(np.float32), mode='F').transform((w, h), Image.AFFINE, mat, Image.BILINEAR)) for ch in range(3)])
        ok = covered(augment_ref.affine(torch.ones(1, 1, h, w), ang, t, sc, sh))
        assert ok.mean() > 0.5 and np.abs(got.numpy() - want)[:, ok].max() < 2e-4, (ang, t, sc, sh)


def check_augment_invariants(lib, dev, size=32, patch=16):
    """Properties the real torchvision ops satisfy (transforms.py:165-170), checked on the HIP sampler: a-8 stays
    "parity unpinned" (no torchvision in the image) -- these pin what can be pinned without it."""
    seed_all(12)
    H, W, S = 48, 64, 6
    img = torch.rand(1, 3, H, W)
    table = R.draw_crop_table(S, size, H, W, 'uniform', 0.4)
    geom = ops.make_geom(H, W, S, size, patch=patch)
    tb = torch.from_numpy(table).to(dev)
    rgb = img[0].to(dev).contiguous()
    plain = ops.sample_fwd(geom, rgb, tb, lib=lib).cpu()                       # normalize()-only path (parity pinned by the goldens)
    start = [[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]]
    ident = augment_ref.perspective_coeffs(start, start)
    assert np.allclose(ident, [1, 0, 0, 0, 1, 0, 0, 0], atol=1e-6)             # endpoints == startpoints -> identity homography
    # (1) nothing drawn: no perspective, no erase, 0 degrees.  torchvision's affine has no identity shortcut: the 0-degree
    #     grid still goes through grid_sample, whose fp32 unnormalisation lands within ~1e-5 px of the pixel centres, so the
    #     result equals the normalize-only cuts to rounding (not bit for bit); the ones-mask is 1 to the same rounding.
    prm0 = [dict(persp=None, erase=None, angle=0.0) for _ in range(S)]
    out0 = ops.sample_fwd(geom, rgb, tb, aug=pack_aug(prm0).to(dev), lib=lib).cpu()
    # (the grid's deviation from the pixel centres grows with the side: ~n x 2^-24 of a pixel, times the image gradient)
    tol = 5e-5 * max(1.0, size / 32.0)
    assert (out0 - plain).abs().max().item() < tol, (out0 - plain).abs().max().item()
    # (2) identity perspective + 0 degrees: same
    prm1 = [dict(persp=ident, erase=None, angle=0.0) for _ in range(S)]
    out1 = ops.sample_fwd(geom, rgb, tb, aug=pack_aug(prm1).to(dev), lib=lib).cpu()
    assert (out1 - plain).abs().max().item() < 2 * tol, (out1 - plain).abs().max().item()
    # (3) erase: the rectangle [i, i+h) x [j, j+w) is 0 BEFORE normalisation, i.e. -mean/std after; everything else untouched
    rect = (3, 5, 7, 9)
    prm2 = [dict(persp=None, erase=rect, angle=None) for _ in range(S)]          # angle None: no rotation stage at all (has_rotation 0)
    out2 = ops.sample_fwd(geom, rgb, tb, aug=pack_aug(prm2).to(dev), lib=lib).cpu()
    zero = R.normalize(torch.zeros(1, 3, 1, 1))
    i, j, eh, ew = rect
    assert torch.allclose(out2[:, :, i:i + eh, j:j + ew], zero.expand(S, 3, eh, ew), atol=1e-6)
    keep = torch.ones(size, size, dtype=torch.bool); keep[i:i + eh, j:j + ew] = False
    assert (out2[:, :, keep] - plain[:, :, keep]).abs().max().item() < 2e-6     # (two kernels, same arithmetic: equal up to FMA contraction)
    # (4) a pure-translation homography (endpoints = startpoints + t) moves the content by +t; pixels whose source falls
    #     outside the cut are exactly the fill (0 before normalisation): the ones-mask blend
    t = (4, 3)
    end = [[x + t[0], y + t[1]] for x, y in start]
    prm3 = [dict(persp=augment_ref.perspective_coeffs(start, end), erase=None, angle=None) for _ in range(S)]
    out3 = ops.sample_fwd(geom, rgb, tb, aug=pack_aug(prm3).to(dev), lib=lib).cpu()
    assert (out3[:, :, t[1]:, t[0]:] - plain[:, :, :size - t[1], :size - t[0]]).abs().max().item() < 2e-4
    assert torch.allclose(out3[:, :, :t[1] - 1, :], zero.expand(S, 3, t[1] - 1, size), atol=1e-6)
    assert torch.allclose(out3[:, :, :, :t[0] - 1], zero.expand(S, 3, size, t[0] - 1), atol=1e-6)
    # (5) +90 degrees: T.functional.affine rotates CLOCKWISE (its inverse matrix [[cos, sin], [-sin, cos]] maps an output pixel
    #     right of the centre to the input pixel above it), exactly a transpose + flip on a square cut
    prm4 = [dict(persp=None, erase=None, angle=90.0) for _ in range(S)]
    out4 = ops.sample_fwd(geom, rgb, tb, aug=pack_aug(prm4).to(dev), lib=lib).cpu()
    assert (out4 - torch.rot90(plain, k=-1, dims=(2, 3))).abs().max().item() < 2e-4


def check_sim_loss(lib, dev, g):
    v1 = torch.from_numpy(g['v1'])
    v2 = torch.from_numpy(g['v2']).contiguous()
    for t in [None, 'mix', 'ang', 'dot']:
        loss, genc = ops.sim_loss(v2.to(dev), v1.to(dev).contiguous(), [1.0], t, lib=lib)
        assert abs(loss.item() - float(g['val_%s' % t])) < 2e-6 * max(1.0, abs(float(g['val_%s' % t]))), t
        assert np.allclose(genc.cpu().numpy(), g['grad_%s' % t], rtol=2e-4, atol=2e-7), t
    tg = torch.randn(2, 64, generator=torch.Generator().manual_seed(4))
    x = v2.clone().requires_grad_(True)
    want = -1.0 * R.sim_func(tg[0:1], x, 'mix') + 0.5 * R.sim_func(tg[1:2], x, 'mix')
    want.backward()
    loss, genc = ops.sim_loss(v2.to(dev), tg.to(dev), [-1.0, 0.5], 'mix', gscale=8.0, lib=lib)
    assert abs(loss.item() - want.item()) < 1e-6
    assert np.allclose(genc.cpu().numpy() / 8.0, x.grad.numpy(), rtol=2e-4, atol=2e-7)


def check_sim_loss_per_cut(lib, dev):
    """reference-image term: per-cut target rows (clip_fft.py:267) mixed with a broadcast prompt, sharded call"""
    g = torch.Generator().manual_seed(6)
    S, Dm = 7, 64
    enc = torch.randn(S, Dm, generator=g)
    txt = torch.randn(1, Dm, generator=g)
    ref = torch.randn(S, Dm, generator=g)
    for t in ['mix', None, 'ang']:
        x = enc.clone().requires_grad_(True)
        want = -1.0 * R.sim_func(txt, x, t) - 0.5 * R.sim_func(ref, x, t)
        want.backward()
        loss, genc = ops.sim_loss(enc.to(dev), txt.to(dev), [-1.0, -0.5], t, per_sample=ref[None].to(dev), lib=lib)
        assert abs(loss.item() - want.item()) < 2e-6, t
        assert np.allclose(genc.cpu().numpy(), x.grad.numpy(), rtol=3e-4, atol=3e-7), t
        # a shard of cuts 2..4 with the global denominator: gradients are the matching rows, losses add up
        l2, g2 = ops.sim_loss(enc[2:5].contiguous().to(dev), txt.to(dev), [-1.0, -0.5], t, denom=S, per_sample=ref[None].to(dev),
                              s_total=S, s_offset=2, lib=lib)
        assert np.allclose(g2.cpu().numpy(), x.grad.numpy()[2:5], rtol=3e-4, atol=3e-7), t


def check_linear_head(lib, dev):
    """aph_linear_head (the aesthetic predictor term) vs torch autograd, accumulating on top of an existing loss / gradient"""
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(9)
    S, Dm = 7, 64
    enc = torch.randn(S, Dm, generator=g).requires_grad_(True)
    w, b = torch.randn(1, Dm, generator=g), 0.3
    want = -0.001 * 25.0 * torch.nn.functional.linear(enc, w, torch.tensor([b])).mean()
    want.backward()
    loss = torch.full((1,), 0.5).to(dev)
    genc0 = torch.randn(S, Dm, generator=g)
    genc = genc0.clone().to(dev)
    d_enc, d_w = enc.detach().to(dev).contiguous(), w.reshape(-1).to(dev).contiguous()        # (held: ops.ptr() keeps no reference)
    L.call('aph_linear_head', ops.ptr(d_enc), S, Dm, ops.ptr(d_w), b, -0.001 * 25.0, float(S), 8.0, ops.ptr(loss), ops.ptr(genc), ops._stream(loss))
    assert abs(loss.item() - 0.5 - want.item()) < 1e-6
    assert np.allclose((genc.cpu() - genc0).numpy() / 8.0, enc.grad.numpy(), rtol=1e-4, atol=2e-7)      # (difference of O(1) f32 values)


def check_adam(lib, dev, n=5000, offset=0):
    """offset > 0: every buffer starts `offset` floats into its allocation (not 16-byte aligned: the scalar kernels)"""
    gen = torch.Generator().manual_seed(0)

    def buf(t):
        if t is None or not offset:
            return t
        big = torch.zeros(t.numel() + offset, device=t.device)
        big[offset:] = t
        return big[offset:]
    for name, kw, dec, ams in [('adam_custom', dict(beta1=0.0), False, False), ('adam', dict(beta1=0.9), False, False),
                               ('adamw', dict(beta1=0.9, weight_decay=0.01), True, False),
                               ('adamw_custom', dict(beta1=0.0, weight_decay=0.01), True, True)]:
        p0 = torch.randn(n, generator=gen)
        q = p0.clone().requires_grad_(True)
        p = buf(p0.to(dev))
        opt = R.make_optimizer([q], name, 0.05)
        m = buf(torch.zeros(n, device=dev)) if kw['beta1'] else None
        v, vm = buf(torch.zeros(n, device=dev)), (buf(torch.zeros(n, device=dev)) if ams else None)
        for step in range(1, 4):
            grad = torch.randn(n, generator=gen) * 0.01
            q.grad = grad.clone()
            opt.step()
            hyper = torch.tensor(ops.adam_hyper(step, 0.05, **kw), dtype=torch.float32).to(dev)
            ops.adam_step(p, buf(grad.to(dev)), m, v, vm, hyper, dec, lib=lib)
            assert (p.cpu() - q.detach()).abs().max().item() < 2e-6, name


def check_gemm(lib, dev, shapes, tile_cfg=0, variants=(1, 0)):
    """variants: MFMA shape of the main loop, 0 = 16x16x32 (the default), 1 = 32x32x16 (kept as a measured alternative) -- both are checked"""
    L = lib if lib is not None else _ffi.lib()
    gen = torch.Generator().manual_seed(0)
    for (M, Nn, K) in shapes:
        A = torch.randn(M, K, generator=gen).half()
        Bt = torch.randn(Nn, K, generator=gen).half()       # asymmetric operands (catches transposes)
        want = A.float() @ Bt.float().T
        for v in variants:
            prev = L.cdll.aph_gemm_set_mfma32(v)
            try:
                C = ops.gemm_f16(A.to(dev), Bt.to(dev), lib=lib, tile_cfg=tile_cfg)
            finally:
                L.cdll.aph_gemm_set_mfma32(prev)
            assert (C.cpu() - want).abs().max().item() < 2e-3 * (K / 64) ** 0.5, (M, Nn, K, v)


TINY = dict(input_resolution=32, patch_size=16, width=256, layers=2, heads=4, output_dim=128)


def check_vit(lib, dev, cfg=TINY, S=3, fwd_tol=3e-3, bwd_tol=2e-2, check_fuse=True, hilo=False):
    w = synthetic_visual_weights(cfg, 3)
    Rr, p = cfg['input_resolution'], cfg['patch_size']
    x = torch.randn(S, 3, Rr, Rr, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
    want = clip_vit_ref.encode_image(w, x, cfg)
    genc = torch.randn(S, cfg['output_dim'], generator=torch.Generator().manual_seed(2)) * 0.01
    (want * genc).sum().backward()
    vit = ops.VitHandle(cfg, w, max_batch=S + 1, lib=lib)
    patches = ops.patchify(x.detach().to(dev).contiguous(), p, lib=lib, hilo=hilo)      # hilo: the split-precision forward (rows [hi | lo])
    if hilo:
        plain = ops.patchify(x.detach().to(dev).contiguous(), p, lib=lib)
        kp = plain.shape[1]
        assert torch.equal(patches[:, :kp], plain)
        xs = plain.float() + patches[:, kp:].float()          # hi + lo reproduces the f32 pixels to ~2^-22
        want_pm = to_patch_major(x.detach(), p).to(dev)
        assert (xs - want_pm).abs().max().item() < 2e-6 * max(want_pm.abs().max().item(), 1.0)
    enc = vit.forward(patches, S, hilo=hilo)
    ferr = (enc.cpu() - want.detach()).abs().max().item() / want.abs().max().item()
    assert ferr < fwd_tol, ferr
    LS = 1024.0
    gp = vit.backward((genc * LS).to(dev).contiguous(), S, out_scale=1.0 / LS)
    gx = ops.unpatchify(gp, S, Rr, p, lib=lib)
    berr = (gx.cpu() - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert berr < bwd_tol, berr
    if not check_fuse or hilo:
        return ferr, berr
    # [r6] the f16-only gradient stream (aph_vit_set_grad_stream_f16: every LayerNorm backward takes its residual from the f16 copy its
    # predecessor wrote and writes no fp32 stream): the same input gradient to within the f16 rounding of the stream, and switching it off
    # again restores the fp32-stream bits
    L = lib if lib is not None else _ffi.lib()
    prev16 = L.call('aph_vit_set_grad_stream_f16', 1)
    try:
        gp16 = vit.backward((genc * LS).to(dev).contiguous(), S, out_scale=1.0 / LS).clone()
    finally:
        L.call('aph_vit_set_grad_stream_f16', prev16)
    gx16 = ops.unpatchify(gp16, S, Rr, p, lib=lib)
    berr16 = (gx16.cpu() - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert berr16 < 2 * bwd_tol and not torch.equal(gp16, gp), (berr16, berr)
    assert torch.equal(vit.backward((genc * LS).to(dev).contiguous(), S, out_scale=1.0 / LS), gp)
    # the fused LayerNorm pairs of the first block (and the unfilled fp32 gradient stream) against the separate kernels: the same
    # arithmetic on the same values, so the results are equal bit for bit
    enc1, gp1 = enc.clone(), gp.clone()
    prev = L.call('aph_vit_set_fuse_ln', 0)
    try:
        enc0 = vit.forward(patches, S).clone()
        gp0 = vit.backward((genc * LS).to(dev).contiguous(), S, out_scale=1.0 / LS).clone()
    finally:
        L.call('aph_vit_set_fuse_ln', prev)
    if dev == 'cpu':
        assert torch.equal(enc0, enc1) and torch.equal(gp0, gp1)
    else:       # (on the GPU the two instantiations may round their multiply-adds differently: tools/exp/ln_fuse_diff.py prints by how much)
        de, dg = (enc0 - enc1).abs().max().item(), (gp0.float() - gp1.float()).abs().max().item()
        print('fused vs separate LayerNorm pairs: max |d enc| %.2e, max |d grad| %.2e' % (de, dg))
        assert de <= fwd_tol * enc1.abs().max().item() and dg <= bwd_tol * gp1.float().abs().max().item(), (de, dg)
    return ferr, berr


def check_rgb_priors(lib, dev):
    """aph_rgb_priors vs torch autograd on the expression of illustrip.py:439-440"""
    from aphantasia_amd import _ffi
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(4)
    for (h, w) in ((37, 53), (64, 200)):
        x = (torch.rand(1, 3, h, w, generator=g) * torch.tensor([0.3, 0.9, 0.6]).view(1, 3, 1, 1)).requires_grad_(True)
        want = abs(x.mean((2, 3)) - 0.45).mean() + abs(x.std((2, 3)) - 0.17).mean()
        want.backward()
        rgb = x.detach().reshape(3, h, w).to(dev).contiguous()
        base = 1e-4 * torch.randn(3, h, w, generator=g)
        grad = base.clone().to(dev)
        loss = torch.full((1,), 0.25, device=dev)
        ws = torch.empty(int(L.cdll.aph_rgb_priors_ws_bytes()) // 8, dtype=torch.float64, device=dev)
        L.call('aph_rgb_priors', ops.ptr(rgb), h, w, 0.45, 0.17, 2.0, ops.ptr(ws), ops.ptr(loss), ops.ptr(grad), ops._stream(rgb))
        assert abs(loss.item() - 0.25 - 2.0 * want.item()) < 1e-6
        err = (grad.cpu() - base - 2.0 * x.grad.reshape(3, h, w)).abs().max().item()
        assert err < 1e-4 * (2.0 * x.grad.abs().max().item() + 1e-4), err


def check_rgb_to_u8(lib, dev):
    """aph_rgb_to_u8 (the saved frame's conversion) vs utils.checkout's arithmetic (utils.py:94-100): np.clip(img ** g * 255, 0, 255).astype(uint8), HWC"""
    from aphantasia_amd import _ffi
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(8)
    for (h, w, gamma) in ((7, 13, 1.0), (33, 50, 1.0), (20, 31, 1.3)):
        x = torch.rand(3, h, w, generator=g) * 1.2 - 0.1
        if gamma != 1.0:
            x = x.clamp_min(0.0)
        out = torch.zeros(h, w, 3, dtype=torch.uint8, device=dev)
        xd = x.to(dev).contiguous()
        L.call('aph_rgb_to_u8', ops.ptr(xd), h, w, float(gamma), ops.ptr(out), ops._stream(xd))
        want = np.clip((x.numpy().astype(np.float32) ** np.float32(gamma)) * np.float32(255), 0, 255).astype(np.uint8).transpose(1, 2, 0)
        d = np.abs(out.cpu().numpy().astype(int) - want.astype(int)).max()
        assert d <= (0 if gamma == 1.0 else 1), (h, w, gamma, d)


def check_rgb_sharp(lib, dev):
    """aph_rgb_sharp vs torch autograd on utils.py:265-268 derivat(img, 'naiv')"""
    from aphantasia_amd import _ffi
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(6)
    for (h, w) in ((19, 33), (64, 130)):
        x = torch.rand(1, 3, h, w, generator=g)
        x[0, 1, 3:6, 4:9] = 0.5                                   # flat patch: ties (sub-gradient 0)
        x.requires_grad_(True)
        dx = torch.mean(torch.abs(x[:, :, :, 1:] - x[:, :, :, :-1]))
        dy = torch.mean(torch.abs(x[:, :, 1:, :] - x[:, :, :-1, :]))
        want = 0.5 * (dx + dy)
        want.backward()
        rgb = x.detach().reshape(3, h, w).to(dev).contiguous()
        grad = torch.zeros(3, h, w, device=dev)
        loss = torch.zeros(1, device=dev)
        ws = torch.empty(int(L.cdll.aph_rgb_priors_ws_bytes()) // 8, dtype=torch.float64, device=dev)
        L.call('aph_rgb_sharp', ops.ptr(rgb), h, w, -0.7, ops.ptr(ws), ops.ptr(loss), ops.ptr(grad), ops._stream(rgb))
        assert abs(loss.item() + 0.7 * want.item()) < 1e-6
        err = (grad.cpu() + 0.7 * x.grad.reshape(3, h, w)).abs().max().item()
        assert err < 1e-5 * x.grad.abs().max().item(), err


def check_frame_affine(lib, dev):
    """aph_frame_affine vs the oracle's restatement of T.functional.affine (bilinear, zero fill, ones-mask)"""
    from aphantasia_amd import transforms
    from oracle import augment_ref
    g = torch.Generator().manual_seed(8)
    for (h, w, args) in ((37, 53, (7.5, (3, -2), 1.1, 4.0)), (48, 80, (0.0, (0, 1), 1.02, 0.0)), (40, 64, (-30.0, (-5, 6), 0.8, (10.0, -3.0)))):
        x = torch.rand(1, 3, h, w, generator=g)
        want = augment_ref.affine(x, *args)
        got = transforms.frame_transform(x.to(dev), (h, w), args[0], args[1], args[2], args[3], lib=lib).cpu()
        err = (got - want).abs().max().item()
        assert err < 2e-5, (err, args)
    x = torch.rand(1, 3, 20, 30, generator=g)
    got = transforms.frame_transform(x.to(dev), (24, 26), 0.0, (0, 0), 1.0, 0.0, lib=lib).cpu()      # identity warp, then pad rows / crop cols
    assert got.shape == (1, 3, 24, 26)
    assert torch.allclose(got[..., 2:22, :], x[..., :, 2:28], atol=1e-5) and float(got[..., :2, :].abs().max()) == 0.0


def check_adam_guard(lib, dev):
    """aph_adam_step_guarded: a NaN / inf anywhere in the gradient skips the step (parameters and moments untouched) and
    counts it; a finite gradient gives exactly aph_adam_step's update"""
    from aphantasia_amd import _ffi
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(3)
    n = 5000
    p0 = torch.randn(n, generator=g); grad = torch.randn(n, generator=g); v0 = torch.rand(n, generator=g)
    hyper = torch.tensor(ops.adam_hyper(3, 0.05, 0.0, 0.999, 1e-8, 0.0, 1.0), dtype=torch.float32).to(dev)
    guard = torch.zeros(2, dtype=torch.int32, device=dev)
    def run(gr, guarded):
        p, v = p0.clone().to(dev), v0.clone().to(dev)
        gr = gr.to(dev)
        if guarded:
            L.call('aph_adam_step_guarded', ops.ptr(p), ops.ptr(gr), None, ops.ptr(v), None, ops.ptr(hyper), 0, n, ops.ptr(guard), ops._stream(p))
        else:
            L.call('aph_adam_step', ops.ptr(p), ops.ptr(gr), None, ops.ptr(v), None, ops.ptr(hyper), 0, n, ops._stream(p))
        return p.cpu(), v.cpu()
    pa, va = run(grad, False)
    pb, vb = run(grad, True)
    assert torch.equal(pa, pb) and torch.equal(va, vb) and guard.cpu().tolist()[0] == 0
    for bad in (float('nan'), float('inf'), -float('inf')):
        gb = grad.clone(); gb[1234] = bad
        pc, vc = run(gb, True)
        assert torch.equal(pc, p0) and torch.equal(vc, v0)
    assert guard.cpu().tolist()[0] == 3


def check_depthwarp(lib, dev, g, sizes=((40, 56), (37, 51), (64, 48))):
    """csrc/depthwarp.hip vs (a) outputs of the reference's own depth/depth.py functions (golden `g`), (b) the oracle at
    other sizes / parameters (reflections, odd sizes, centre outside the frame)"""
    from aphantasia_amd import depthwarp as DW
    from oracle import depth_ref
    t = lambda k: torch.from_numpy(g[k])
    img_t, img, dep = t('img_t'), t('img'), t('dep')
    H, W = img.shape[-2:]
    tol = lambda want: 2e-5 * max(1.0, want.abs().max().item())
    got = DW.grid_warp(img_t.to(dev), dep.to(dev), H, W, 0.3, [0.1, -0.2], 0.5, lib=lib).cpu()
    assert (got - t('warp_a')).abs().max().item() < tol(t('warp_a'))
    got = DW.grid_warp(img_t.to(dev), dep.to(dev), H, W, 4.0, [1.5, 0.7], 0.2, dlens=0.3, lib=lib).cpu()
    assert (got - t('warp_b')).abs().max().item() < 5 * tol(t('warp_b'))        # strength 4: coordinates far outside, several reflections
    got = DW.triangle_blur(img.to(dev), 5, 2, lib=lib).cpu()
    assert (got - t('blur')).abs().max().item() < 2e-6
    got = DW.resize(img.to(dev), (28, 42), lib=lib).cpu()
    assert (got - t('resize_dn')).abs().max().item() < 5e-6
    got = DW.resize(dep[None].to(dev), (70, 75), lib=lib).cpu()
    assert (got - t('resize_up')).abs().max().item() < 5e-6
    dev_depth = lambda image: depth_ref.toy_depth(image.cpu()).to(dev)      # the estimator is the caller's business
    got = DW.depthwarp(img_t.to(dev), img.to(dev), dev_depth, 0.4, [0.2, -0.1], 0.6, lib=lib).cpu()
    assert (got - t('depthwarp')).abs().max().item() < tol(t('depthwarp'))
    gen = torch.Generator().manual_seed(4)
    for (h, w) in sizes:
        x = torch.randn(1, 3, h, w, generator=gen)
        d = torch.rand(1, h, w, generator=gen)
        if h * w > 100000:
            # A bilinear sample moves by (image gradient) x (coordinate error), and fp32 grid coordinates carry ~4e-5 px of rounding at
            # 1280 px (torch's own linspace differs by 1 ulp between AVX2 and AVX-512 hosts): white noise at full size measures
            # that, not the kernel.  Full-size frames are checked on a smooth picture (what the loop warps), noise at small sizes.
            x = torch.nn.functional.interpolate(torch.randn(1, 3, h // 16, w // 16, generator=gen), (h, w), mode='bicubic', align_corners=False)
            d = torch.nn.functional.interpolate(torch.rand(1, 1, h // 16, w // 16, generator=gen), (h, w), mode='bilinear')[0].clamp(0, 1)
        cases = ((0.25, (0.0, 0.0), 0.5, 0.05), (-1.5, (-2.0, 0.3), 0.9, 0.5), (0.0, (0.3, 0.3), 0.5, 0.05))
        if h * w > 100000:      # (an extreme warp folds the picture: pass 2 then samples gradients of several units per pixel -- small sizes only)
            cases = ((0.25, (0.0, 0.0), 0.5, 0.05), (0.6, (0.3, -0.2), 0.7, 0.1), (0.0, (0.3, 0.3), 0.5, 0.05))
        for (strength, centre, mid, dl) in cases:
            want = depth_ref.grid_warp(x, d, h, w, strength, centre, mid, dl)
            got = DW.grid_warp(x.to(dev), d.to(dev), h, w, strength, centre, mid, dl, lib=lib).cpu()
            err = (got - want).abs()
            rel = 3e-4 if h * w > 100000 else 1e-4          # full size: ~1e-4 px of fp32 coordinate rounding on either side
            assert err.max().item() < rel * max(1.0, abs(strength)) * want.abs().max().item(), (h, w, strength, err.max().item())
            assert err.mean().item() < 1e-5 * want.abs().max().item(), (h, w, strength, err.mean().item())
        for (k, p, mix) in ((5, 2.0, 0.5), (3, 1.0, 1.0), (7, 1.5, 0.25)):
            want = torch.lerp(x, depth_ref.triangle_blur(x, k, p), mix)
            got = DW.triangle_blur(x.to(dev), k, p, mix=mix, lib=lib).cpu()
            assert (got - want).abs().max().item() < 3e-6, (h, w, k)
        for size in ((h * 2 + 1, w * 3), (h // 2, w // 3 + 1), (1, 1), (h, w)):
            want = depth_ref.resize(x, size)
            got = DW.resize(x.to(dev), size, lib=lib).cpu()
            assert (got - want).abs().max().item() < 1e-5, (h, w, size)
        want = torch.flip(x, [-1]) * x
        got = DW.flip_w(x.to(dev), mul=x.to(dev), lib=lib).cpu()
        assert torch.equal(got, want)
    # zero strength: both passes are the identity grid -> the image itself (to interpolation rounding)
    x = torch.randn(1, 3, 33, 47, generator=gen)
    got = DW.grid_warp(x.to(dev), torch.rand(1, 33, 47, generator=gen).to(dev), 33, 47, 0.0, [0.4, 0.1], 0.5, lib=lib).cpu()
    assert (got - x).abs().max().item() < 2e-5


def attention_ref(qkv, S, T, heads):
    """plain torch fp32 multi-head attention on the packed [S*T, 3*heads*64] activations (clip/model.py's nn.MultiheadAttention core:
    softmax(q k^T / sqrt(64)) v per head); returns att [S*T, heads*64] and the log-sum-exp of the scaled scores [S, heads, T]"""
    D = heads * 64
    x = qkv.float().reshape(S, T, 3, heads, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))            # [S, heads, T, 64]
    sc = q @ k.transpose(-1, -2) * 0.125
    att = torch.softmax(sc, dim=-1) @ v
    return att.permute(0, 2, 1, 3).reshape(S * T, D), torch.logsumexp(sc, dim=-1)


def check_attention(lib, dev, S, T, heads, seed=0):
    """the attention kernels alone (aph_attn_test) vs torch fp32 autograd on the same f16-rounded inputs"""
    L = lib if lib is not None else _ffi.lib()
    g = torch.Generator().manual_seed(seed)
    D = heads * 64
    qkv = (torch.randn(S * T, 3 * D, generator=g) * 1.5).half()
    datt = torch.randn(S * T, D, generator=g).half()
    ref_in = qkv.float().requires_grad_(True)
    want_att, want_lse = attention_ref(ref_in, S, T, heads)
    (want_att * datt.float()).sum().backward()
    q_d, d_d = qkv.to(dev).contiguous(), datt.to(dev).contiguous()
    att = torch.empty(S * T, D, dtype=torch.float16, device=dev)
    lse = torch.empty(S * heads * T, dtype=torch.float32, device=dev)
    dqkv = torch.zeros(S * T, 3 * D, dtype=torch.float16, device=dev)
    delta = torch.empty(S * heads * T, dtype=torch.float32, device=dev)
    st = ops._stream(q_d)
    L.call('aph_attn_test', ops.ptr(q_d), ops.ptr(att), ops.ptr(lse), None, None, None, S, T, heads, 0, st)
    assert (att.float().cpu() - want_att.detach()).abs().max().item() < 4e-3 * want_att.abs().max().item()
    assert (lse.cpu().reshape(S, heads, T) - want_lse.detach()).abs().max().item() < 2e-3
    L.call('aph_attn_test', ops.ptr(q_d), ops.ptr(att), ops.ptr(lse), ops.ptr(d_d), ops.ptr(delta), ops.ptr(dqkv), S, T, heads, 1, st)
    err = (dqkv.float().cpu() - ref_in.grad).abs().max().item()
    assert err < 8e-3 * ref_in.grad.abs().max().item(), (S, T, heads, err, ref_in.grad.abs().max().item())
    return att, dqkv


def check_sampler_fuzz(lib, dev, seed, n, max_hw=(90, 120)):
    """Random ragged geometries (image sides from just above the cut size, all --align modes, every output layout, with and without the
    -tf fast chain, 1..8 cuts): forward and adjoint vs the oracle.  Found the cull bug of images under 16 pixels on a side."""
    rng = np.random.default_rng(seed)
    for it in range(n):
        size = int(rng.choice([8, 16, 32]))
        patch = int(rng.choice([4, 8]))
        H = int(rng.integers(size + 1, max_hw[0])); W = int(rng.integers(size + 1, max_hw[1]))
        S = int(rng.integers(1, 9)); align = str(rng.choice(['uniform', 'central', 'overscan', 'overmax']))
        mode = int(rng.choice([_ffi.APH_OUT_NCHW_RAW, _ffi.APH_OUT_NCHW_NORM, _ffi.APH_OUT_PATCH_F16]))
        aug = bool(rng.integers(0, 2))
        sd = int(rng.integers(0, 10000))
        seed_all(sd)
        case = dict(size=size, patch=patch, H=H, W=W, S=S, align=align, mode=mode, aug=aug, seed=sd)
        img = torch.rand(1, 3, H, W).requires_grad_(True)
        table = R.draw_crop_table(S, size, H, W, align, 0.4)
        prm = [augment_ref.draw_fast_params(size) for _ in range(S)] if aug else None
        tf = None if mode == _ffi.APH_OUT_NCHW_RAW else R.normalize
        if aug:
            ident = lambda x: x
            cuts = R.slice_imgs(img, table, size, align, transform=tf, per_cut=lambda c, cut: augment_ref.apply_fast(cut, prm[c], tf if tf is not None else ident))
        else:
            cuts = R.slice_imgs(img, table, size, align, transform=tf)
        gout = torch.randn_like(cuts)
        (cuts * gout).sum().backward()
        geom = ops.make_geom(H, W, S, size, patch=patch, align=align)
        tb = torch.from_numpy(table).to(dev)
        augt = pack_aug(prm).to(dev) if aug else None
        got = ops.sample_fwd(geom, img.detach()[0].to(dev).contiguous(), tb, aug=augt, out_mode=mode, lib=lib).cpu()
        want = cuts.detach()
        if mode == _ffi.APH_OUT_PATCH_F16:
            want, got = to_patch_major(want, patch).half().float(), got.float()
        e1 = (got - want).abs().max().item()
        assert e1 < (4e-3 if mode == _ffi.APH_OUT_PATCH_F16 else 1e-4), (case, e1)
        gin = to_patch_major(gout, patch) if mode == _ffi.APH_OUT_PATCH_F16 else gout
        gb = ops.sample_bwd(geom, gin.to(dev).contiguous(), tb, aug=augt, out_mode=mode, gscale=1.0, lib=lib).cpu()
        e2 = (gb - img.grad[0]).abs().max().item() / max(img.grad.abs().max().item(), 1e-9)
        assert e2 < 2e-4, (case, e2)
