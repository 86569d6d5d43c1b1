"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by running the REFERENCE'S OWN
functions (imported in place from /root/reference through oracle/shim.py) on seeded
inputs.  Run in the build container only:  python -m oracle.make_goldens
The fixtures are small (a few hundred KB) and are committed; the GPU box never
needs /root/reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shim, clip_vit_ref  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
TINY_VIT = dict(input_resolution=16, patch_size=8, width=64, layers=2, heads=2, output_dim=32)


def tiny_weights(seed=1):
    from aphantasia_amd.weights import synthetic_visual_weights
    return synthetic_visual_weights(TINY_VIT, seed)


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def gold_synth(ref, h, w, name, colors=1.8, decay=1.5, contrast=1.1):
    seed_all(0)
    params, image_f, _ = ref.fft_image([1, 3, h, w], 0.07, decay, None)
    rgb_f = ref.to_valid_rgb(image_f, colors=colors)
    raw = image_f()                       # contrast 1
    rgb = rgb_f(contrast=contrast)
    gw = torch.randn(1, 3, h, w, generator=torch.Generator().manual_seed(5))
    (rgb * gw).sum().backward()
    np.savez_compressed(os.path.join(OUT, name), h=h, w=w, colors=colors, decay=decay, contrast=contrast,
                        params=params[0].detach().numpy(), raw=raw.detach().numpy(), rgb=rgb.detach().numpy(),
                        gw=gw.numpy(), grad=params[0].grad.numpy())


def gold_slice(ref, name):
    out = {}
    img = torch.rand(1, 3, 48, 80, generator=torch.Generator().manual_seed(3))
    out['img'] = img.numpy()
    for align in ['uniform', 'central', 'overscan', 'overmax']:
        seed_all(7)
        cuts = ref.slice_imgs([img], 6, 16, ref.normalize(), align, 0.4)[0]
        out['cuts_' + align] = cuts.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def gold_sim(ref, name):
    g = torch.Generator().manual_seed(11)
    v1 = torch.randn(1, 64, generator=g)
    v2 = torch.randn(5, 64, generator=g)
    out = dict(v1=v1.numpy(), v2=v2.numpy())
    for t in [None, 'mix', 'ang', 'dot']:
        x = v2.clone().requires_grad_(True)
        val = ref.sim_func(v1, x, t)
        val.backward()
        out['val_%s' % t] = val.detach().numpy()
        out['grad_%s' % t] = x.grad.numpy()
    out['val_spher'] = ref.sim_func(v1, v2, 'spher').numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def gold_run(ref, name, steps=6):
    """clip_fft.py train(i) with the reference's own parameteriser/sampler/loss and
    torch.optim.Adam(lr .05, betas (0,.999)); ViT = restated tiny model (seeded)."""
    w = tiny_weights(1)
    enc = lambda x: clip_vit_ref.encode_image(w, x, TINY_VIT)
    seed_all(0)
    h, wd = 40, 56
    params, image_f, _ = ref.fft_image([1, 3, h, wd], 0.07, 1.5, None)
    rgb_f = ref.to_valid_rgb(image_f, colors=1.8)
    opt = torch.optim.Adam(params, 0.05, betas=(.0, .999))
    target = torch.randn(1, 32, generator=torch.Generator().manual_seed(2))
    p0 = params[0].detach().clone().numpy()
    losses = []
    seed_all(123)
    for i in range(steps):
        img = rgb_f()
        cuts = ref.slice_imgs([img], 5, 16, ref.normalize(), 'uniform', 0.4)[0]
        loss = -1.0 * ref.sim_func(target, enc(cuts), 'mix')
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    with torch.no_grad():
        final = rgb_f(contrast=1.1).numpy()
    np.savez_compressed(os.path.join(OUT, name), h=h, w=wd, params0=p0, target=target.numpy(),
                        losses=np.array(losses), params_final=params[0].detach().numpy(), final=final)


def gold_resume(ref, name):
    """resume-from-image helpers of the reference (image.py:179-220): inv_sigmoid, un_rgb, un_spectrum, img2fft, and
    pixel_image's `3.3 * un_rgb(img, colors=2.)` init"""
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (30, 44, 3), generator=g).numpy().astype(np.uint8)
    spec = torch.randn(1, 3, 30, 23, 2, generator=g)
    x = torch.rand(1, 3, 5, 7, generator=g)
    out = dict(img=img, spec=spec.numpy(), x=x.numpy(),
               inv_sigmoid=ref.image.inv_sigmoid(x).numpy(),
               un_rgb_c15=ref.image.un_rgb(img, colors=1.5).numpy(),
               un_rgb_tensor=ref.image.un_rgb(x, colors=1.0).numpy(),
               un_spectrum=ref.image.un_spectrum(spec.clone(), 1.5).numpy(),
               img2fft=ref.image.img2fft(img, 1.5, 1.5).numpy())
    np.savez_compressed(os.path.join(OUT, name), **out)


def gold_depth(name):
    """the reference's own depth/depth.py grid_warp + depthwarp (depth.py:44-84) and illustrip.py's depth_transform inputs;
    the estimator is oracle.depth_ref.toy_depth (Depth-Anything's weights are not in this image)"""
    from oracle import depth_ref
    d = shim.load_reference_depth()
    g = torch.Generator().manual_seed(21)
    H, W = 40, 56
    img_t = torch.randn(1, 3, H, W, generator=g) * 0.7
    img = torch.rand(1, 3, H, W, generator=g)
    dep = torch.rand(1, H, W, generator=g)
    out = dict(img_t=img_t.numpy(), img=img.numpy(), dep=dep.numpy())
    out['warp_a'] = d.grid_warp(img_t, dep, H, W, 0.3, torch.as_tensor([0.1, -0.2]), 0.5).numpy()
    out['warp_b'] = d.grid_warp(img_t, dep, H, W, 4.0, torch.as_tensor([1.5, 0.7]), 0.2, dlens=0.3).numpy()   # reflections
    out['blur'] = d.triangle_blur(img, 5, 2).numpy()
    out['resize_dn'] = d.resize(img, (28, 42)).numpy()
    out['resize_up'] = d.resize(dep[None], (70, 75)).numpy()
    # depthwarp with a small estimator resolution through a patched `res` is not possible (hard-coded 518, depth.py:71):
    # run it as written -- a 40x56 frame gives a 518x714 estimator input
    out['depthwarp'] = d.depthwarp(img_t, img, depth_ref.toy_depth, 0.4, [0.2, -0.1], 0.6).numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = shim.load_reference()
    gold_synth(ref, 48, 80, 'synth_48x80.npz')
    gold_synth(ref, 45, 63, 'synth_45x63.npz', colors=1.0, decay=1.0, contrast=1.0)   # odd sizes
    gold_slice(ref, 'slice_48x80.npz')
    gold_sim(ref, 'sim.npz')
    gold_depth('depthwarp_40x56.npz')
    gold_run(ref, 'run_40x56.npz')
    gold_resume(ref, 'resume_img.npz')
    print('goldens written to', OUT)


if __name__ == '__main__':
    main()
