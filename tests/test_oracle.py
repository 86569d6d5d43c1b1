"""CPU: the oracle restatement (oracle/) against (a) fixtures produced by the reference's
own functions (tests/golden, made by oracle/make_goldens.py) and (b) the survey's KAT table."""
import numpy as np
import pytest
import torch

from oracle import reference_path as R
from oracle import clip_vit_ref
from aphantasia_amd.weights import synthetic_visual_weights, visual_config


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def test_kat_fft_params_and_rgb():
    # SURVEY.md section 4 KAT rows 1-2 (values recorded from the reference itself)
    seed_all(0)
    p = R.fft_params_init([1, 3, 224, 224])
    assert np.allclose(p[0, 0, 0, 0].numpy(), [-0.01125840, -0.01152360], atol=1e-8)
    assert abs(p.sum().item() - (-2.690636396)) < 1e-4
    scale = R.fft_scale(224, 224, 1.5)
    img = R.synth_fft(p, scale, 224, 224, R.colcorr_t(1.8))
    assert abs(img.mean().item() - 0.49129492) < 1e-6
    assert abs(img.std().item() - 0.15267497) < 1e-6
    assert np.allclose(img[0, :, 0, 0].numpy(), [0.49314794, 0.45017162, 0.58190310], atol=1e-6)
    assert np.allclose(img[0, :, 100, 57].numpy(), [0.49892253, 0.43991053, 0.51003355], atol=1e-6)


def test_kat_colcorr_and_scale():
    cc = R.colcorr_t(1.8).numpy()
    want = [[0.56282848, 0.58447582, 0.58447582], [0.35068548, 0, -0.35068548], [0.07793011, -0.19482526, 0.11689515]]
    assert np.allclose(cc, want, atol=1e-7)
    s = R.fft_scale(720, 1280, 1.5)
    assert tuple(s.shape) == (720, 641)
    assert abs(s.min().item() - 1614.5211) < 1e-2 and abs(s.max().item() - 5495360.66) < 1.0


def test_kat_sim_func():
    v1 = torch.sin(torch.arange(512.))[None]
    v2 = torch.cos(torch.arange(1536.)).view(3, 512)
    want = {None: 0.02639321, 'mix': -0.27291375, 'ang': 0.50846374, 'dot': 14.82776833}
    for t, val in want.items():
        assert abs(R.sim_func(v1, v2, t).item() - val) < 2e-5 * max(1, abs(val)), t
    assert np.allclose(R.sim_func(v1, v2, 'spher').numpy(), [1.23344505, 1.36218655, 0.99605185], atol=1e-5)


@pytest.mark.parametrize('name', ['synth_48x80.npz', 'synth_45x63.npz'])
def test_synth_vs_reference_golden(golden, name):
    g = golden(name)
    h, w = int(g['h']), int(g['w'])
    p = torch.from_numpy(g['params']).requires_grad_(True)
    scale = R.fft_scale(h, w, float(g['decay']))
    cc = R.colcorr_t(float(g['colors']))
    raw = R.std_normalise(R.fft_image_raw(p, scale, h, w))
    assert np.allclose(raw.detach().numpy(), g['raw'], atol=1e-6)
    rgb = R.synth_fft(p, scale, h, w, cc, float(g['contrast']))
    assert np.allclose(rgb.detach().numpy(), g['rgb'], atol=1e-6)
    (rgb * torch.from_numpy(g['gw'])).sum().backward()
    assert np.allclose(p.grad.numpy(), g['grad'], rtol=1e-5, atol=1e-5 * np.abs(g['grad']).max())


@pytest.mark.parametrize('align', ['uniform', 'central', 'overscan', 'overmax'])
def test_slice_vs_reference_golden(golden, align):
    g = golden('slice_48x80.npz')
    img = torch.from_numpy(g['img'])
    seed_all(7)
    table = R.draw_crop_table(6, 16, 48, 80, align, 0.4)
    cuts = R.slice_imgs(img, table, 16, align)
    assert np.array_equal(cuts.numpy(), g['cuts_' + align])


def test_bicubic_matrix_matches_interpolate():
    # the separable 4-tap restatement the HIP sampler implements == F.interpolate
    x = torch.rand(1, 3, 301, 301, generator=torch.Generator().manual_seed(0))
    for n in (225, 301):
        ref = R.crop_resize(x, n, 0, 0, 224)
        Wm = R.bicubic_matrix(n, 224)
        got = torch.einsum('in,bcnm,jm->bcij', Wm, x[:, :, :n, :n], Wm)
        assert (ref - got).abs().max().item() < 5e-6


def test_sim_vs_reference_golden(golden):
    g = golden('sim.npz')
    v1 = torch.from_numpy(g['v1'])
    for t in [None, 'mix', 'ang', 'dot']:
        x = torch.from_numpy(g['v2']).requires_grad_(True)
        val = R.sim_func(v1, x, t)
        val.backward()
        assert np.allclose(val.item(), g['val_%s' % t], rtol=1e-6)
        assert np.allclose(x.grad.numpy(), g['grad_%s' % t], rtol=1e-5, atol=1e-7)


def test_adam_explicit_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    for name, kw in [('adam_custom', dict(beta1=0.0)), ('adam', dict(beta1=0.9)),
                     ('adamw', dict(beta1=0.9, weight_decay=0.01, decoupled=True)),
                     ('adamw_custom', dict(beta1=0.0, weight_decay=0.01, decoupled=True, amsgrad=True))]:
        p = torch.randn(1000, generator=g)
        q = p.clone().requires_grad_(True)
        opt = R.make_optimizer([q], name, 0.05)
        m, v, vm = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        for step in range(1, 5):
            grad = torch.randn(1000, generator=g) * 10 ** float(torch.randn(1, generator=g))
            q.grad = grad.clone()
            opt.step()
            R.adam_step(p, grad, m, v, vm, step, 0.05, **kw)
            assert (p - q.detach()).abs().max().item() < 2e-6, name


def test_run_vs_reference_golden(golden):
    """oracle ReferenceRun == the reference's own functions, free-running 6 Adam steps."""
    from oracle.make_goldens import TINY_VIT, tiny_weights
    g = golden('run_40x56.npz')
    w = tiny_weights(1)
    run = R.ReferenceRun(int(g['h']), int(g['w']), lambda x: clip_vit_ref.encode_image(w, x, TINY_VIT),
                         [(torch.from_numpy(g['target']), 1.0)], size=16, params=torch.from_numpy(g['params0']))
    seed_all(123)
    losses = []
    for i in range(len(g['losses'])):
        table = R.draw_crop_table(5, 16, run.h, run.w, 'uniform', 0.4)
        losses.append(run.step(table))
    assert np.allclose(losses, g['losses'], atol=1e-6)
    assert np.allclose(run.params.detach().numpy(), g['params_final'], atol=1e-5)
    with torch.no_grad():
        assert np.allclose(run.image(1.1).numpy(), g['final'], atol=1e-5)


@pytest.mark.parametrize('name', ['ViT-B/32', 'ViT-B/16'])
def test_vit_restatement_vs_hf(name):
    """openai/CLIP restatement == HF transformers CLIP vision tower (output and input-grad)."""
    cfg = visual_config(name)
    w = synthetic_visual_weights(cfg, 1)
    hf = clip_vit_ref.hf_model(w, cfg)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(4)).requires_grad_(True)
    a = clip_vit_ref.encode_image(w, x, cfg)
    ga, = torch.autograd.grad(a.square().sum(), x)
    x2 = x.detach().clone().requires_grad_(True)
    b = hf(pixel_values=x2).image_embeds
    gb, = torch.autograd.grad(b.square().sum(), x2)
    assert (a - b).abs().max().item() < 2e-5 * b.abs().max().item() + 1e-5
    assert (ga - gb).abs().max().item() < 1e-4 * gb.abs().max().item()


# ---------------------------------------------------------------------------------------------------------------------
# a-8 (transforms_fast): torchvision is absent, so oracle/augment_ref.py stays "parity unpinned"; these are the properties
# of the real ops (transforms.py:165-170; torchvision RandomPerspective / RandomErasing / functional.affine) that can be
# checked without it
# ---------------------------------------------------------------------------------------------------------------------
def test_augment_oracle_invariants():
    from oracle import augment_ref as A
    n = 24
    x = torch.rand(1, 3, n, n, generator=torch.Generator().manual_seed(3))
    start = [[0, 0], [n - 1, 0], [n - 1, n - 1], [0, n - 1]]
    assert np.allclose(A.perspective_coeffs(start, start), [1, 0, 0, 0, 1, 0, 0, 0], atol=1e-6)
    assert (A.perspective(x, A.perspective_coeffs(start, start)) - x).abs().max().item() < 1e-5     # identity homography
    assert (A.rotate(x, 0.0) - x).abs().max().item() < 1e-5                                          # 0 degrees (no shortcut upstream: grid_sample of the identity grid)
    assert (A.rotate(x, 90.0) - torch.rot90(x, k=-1, dims=(2, 3))).abs().max().item() < 1e-5          # `affine`: clockwise
    assert (A.rotate(x, -90.0) - torch.rot90(x, k=1, dims=(2, 3))).abs().max().item() < 1e-5
    assert (A.rotate(A.rotate(x, 180.0), 180.0) - x).abs().max().item() < 1e-5
    # translation homography: content moves by +t, the uncovered border is exactly the fill value 0
    t = (3, 2)
    y = A.perspective(x, A.perspective_coeffs(start, [[a + t[0], b + t[1]] for a, b in start]))
    assert (y[:, :, t[1]:, t[0]:] - x[:, :, :n - t[1], :n - t[0]]).abs().max().item() < 1e-4
    assert float(y[:, :, :t[1] - 1, :].abs().max()) == 0.0 and float(y[:, :, :, :t[0] - 1].abs().max()) == 0.0
    # a 30-degree rotation leaves the corners (outside the rotated frame) exactly 0 and the centre value untouched in the mean
    r = A.rotate(torch.ones(1, 3, n, n), 30.0)
    assert float(r[0, :, 0, 0].abs().max()) == 0.0 and float(r[0, :, -1, -1].abs().max()) == 0.0
    assert abs(float(r[0, 0, n // 2, n // 2]) - 1.0) < 1e-6


def test_augment_geometry_vs_pil_float_transforms():
    """a-8 / f-1 without torchvision: the restated tensor-path maths (oracle/augment_ref.py: perspective coefficients, inverse affine
    matrix, base grids, bilinear grid_sample) against an INDEPENDENT implementation that IS executable here -- Pillow's own float ('F' mode)
    `Image.transform(PERSPECTIVE / AFFINE, BILINEAR)`, i.e. what torchvision's PIL backend calls with the same coefficient helpers
    (`F_pil.perspective` / `F_pil.affine`; the affine matrix there is taken about the image centre).  Interior pixels agree to 1e-5: geometry,
    direction of rotation, centre convention, half-pixel offsets and the interpolation are pinned; the border band differs by construction (PIL
    fills, the tensor path blends with a sampled ones-mask) and stays covered by the ones-mask invariants only."""
    Image = pytest.importorskip('PIL.Image')
    from oracle import augment_ref as A
    g = torch.Generator().manual_seed(0)
    h, w = 64, 80
    x = torch.rand(1, 1, h, w, generator=g)
    img = Image.fromarray(x[0, 0].numpy().astype(np.float32), mode='F')
    def covered(warp_ones):        # whole bilinear footprint inside the source, eroded by one pixel (elsewhere Pillow fills, the tensor path blends)
        m = -torch.nn.functional.max_pool2d(-(warp_ones > 1 - 1e-6).float(), 3, 1, 1)
        return m[0, 0].bool().numpy()
    ones = torch.ones(1, 1, h, w)
    start = [[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]]
    for end in ([[5, 7], [70, 2], [75, 60], [3, 55]], [[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], [[9, 1], [78, 8], [70, 62], [1, 50]]):
        co = A.perspective_coeffs(start, end)
        ref = A.perspective(x, co)[0, 0].numpy()
        pil = np.asarray(img.transform(img.size, Image.PERSPECTIVE, co, Image.BILINEAR))
        ok = covered(A.perspective(ones, co))
        assert ok.mean() > 0.5 and np.abs(ref - pil)[ok].max() < 3e-5, end
    for (ang, t, sc, sh) in ((30.0, [0, 0], 1.0, 0.0), (-17.0, [0, 0], 1.0, 0.0), (0.8, [0.0, 10.0], 1.012, 0.4), (-2.5, [7.0, -3.0], 0.97, -1.2)):
        a, b, c0, d, e, f0 = A.inverse_affine_matrix(ang, t, sc, sh)            # about the centre; PIL wants it in pixel coordinates
        cx, cy = w * 0.5, h * 0.5
        mat = [a, b, c0 + cx - a * cx - b * cy, d, e, f0 + cy - d * cx - e * cy]
        pil = np.asarray(img.transform(img.size, Image.AFFINE, mat, Image.BILINEAR))
        ok = covered(A.affine(ones, ang, t, sc, sh))
        assert ok.mean() > 0.5 and np.abs(A.affine(x, ang, t, sc, sh)[0, 0].numpy() - pil)[ok].max() < 3e-5, (ang, t, sc, sh)
        if t == [0, 0] and sc == 1.0 and sh == 0.0:
            assert np.abs(A.rotate(x, ang)[0, 0].numpy() - pil)[ok].max() < 3e-5, ang        # random_rotate_fast: the same map


def test_torchvision_fixture():
    """a-8 / f-1: the restated torchvision maths against torchvision's own outputs (tests/golden/tf_fast_224.npz, when someone with
    torchvision has generated it: oracle/make_tv_fixture.py); skipped with the reason otherwise"""
    import kernel_checks as K
    K.check_oracle_vs_tv_fixture(K.tv_fixture_or_skip())


def test_augment_draw_semantics():
    """RandomErasing.get_params (scale (0.02, 0.33), log-uniform ratio (0.3, 3.3), value 0) and RandomPerspective.get_params
    (distortion 0.33) bounds; the product's host draws consume the same stream as the oracle's restatement"""
    from oracle import augment_ref as A
    from aphantasia_amd import transforms as T
    size = 224
    seed_all(5)
    n_e = n_p = 0
    for _ in range(400):
        rect = A.erase_get_params(size, size)
        if rect is not None:
            i, j, h, w = rect
            n_e += 1
            assert 0 <= i and i + h <= size and 0 <= j and j + w <= size and h < size and w < size
            assert 0.02 * 0.9 <= h * w / size ** 2 <= 0.33 * 1.1 and 0.3 * 0.9 <= h / w <= 3.3 * 1.1
        sp, ep = A.perspective_get_params(size, size, 0.33)
        n_p += 1
        d = int(0.33 * (size // 2))
        assert sp == [[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]]
        for (x0, y0), (x1, y1) in zip(sp, ep):
            assert abs(x1 - x0) <= d and abs(y1 - y0) <= d
    assert n_e > 350
    # same seed -> same parameters from the product's host code (aphantasia_amd/transforms.py) and the oracle's restatement
    seed_all(77)
    want = [A.draw_fast_params(size) for _ in range(50)]
    seed_all(77)
    got = T.finish_draws([T.transforms_fast.draw(size) for _ in range(50)])       # (the perspective systems are solved in one batch)
    for a, b in zip(want, got):
        assert a['erase'] == b['erase'] and a['angle'] == b['angle'] and (a['persp'] is None) == (b['persp'] is None)
        if a['persp'] is not None:
            assert np.allclose(a['persp'], b['persp'], rtol=1e-5, atol=1e-7)
    assert sum(w['persp'] is not None for w in want) > 3 and sum(w['erase'] is not None for w in want) > 3


def test_depthwarp_oracle_vs_reference_golden(golden):
    """oracle/depth_ref.py == the reference's own depth/depth.py grid_warp / depthwarp / resize and utils.triangle_blur"""
    from oracle import depth_ref as D
    g = golden('depthwarp_40x56.npz')
    t = lambda k: torch.from_numpy(g[k])
    img_t, img, dep = t('img_t'), t('img'), t('dep')
    assert torch.equal(D.grid_warp(img_t, dep, 40, 56, 0.3, [0.1, -0.2], 0.5), t('warp_a'))
    assert torch.equal(D.grid_warp(img_t, dep, 40, 56, 4.0, [1.5, 0.7], 0.2, dlens=0.3), t('warp_b'))
    assert torch.equal(D.triangle_blur(img, 5, 2), t('blur'))
    assert torch.equal(D.resize(img, (28, 42)), t('resize_dn'))
    assert torch.equal(D.resize(dep[None], (70, 75)), t('resize_up'))
    assert torch.equal(D.depthwarp(img_t, img, D.toy_depth, 0.4, [0.2, -0.1], 0.6), t('depthwarp'))


@pytest.mark.parametrize('name', ['c2_s32_stress', 'c2_s48_stress'])
def test_loss_curve_fixture_first_step_reproduces(name):
    """The committed free-running trajectories (tests/golden/loss_curve_*.npz, oracle/make_loss_curves.py) belong to THIS oracle: the
    step-0 loss of a fresh ReferenceRun with the generator's seeds is the fixture's first entry (fp32 sums: thread-count noise only)."""
    import os
    import numpy as np
    import torch
    from oracle import clip_vit_ref, make_loss_curves as M, reference_path as R
    c = M.CONFIGS[name]
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loss_curve_%s.npz' % name))
    assert len(fx['loss']) == c['steps'] and ('%d cuts' % c['S']) in str(fx['meta'])
    cfg, wts = M.weights_of(c['weights'])
    M.seed_all(0)
    p0 = R.fft_params_init([1, 3, c['h'], c['w']])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    ref = R.ReferenceRun(c['h'], c['w'], lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
    M.seed_all(9)
    table = R.draw_crop_table(c['S'], 224, c['h'], c['w'], 'uniform', 0.4)
    with torch.no_grad():
        got = float(ref.loss(table))
    assert abs(got - float(fx['loss'][0])) < 5e-6, (got, float(fx['loss'][0]))
