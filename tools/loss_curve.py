"""Free-running loss-vs-step of the HIP path against the fp32 CPU oracle (north_star: "loss-vs-step curve matching CPU reference
to 1e-3") at BASELINE configs[1]'s real workload.  Both sides take their own Adam steps from the same init with the same crop
tables; the oracle's trajectory comes from the committed fixture tests/golden/loss_curve_<name>.npz (oracle/make_loss_curves.py:
8-18 s of host CPU per 200-cut step, generated once instead of on GPU-box minutes).

    python tools/loss_curve.py <name> [csv-out] [final-image-out.npy]       name in c2_s200 | c2_s200_200 | c2_s32 | c2_s32_stress | c2_s190_fast
    python tools/loss_curve.py --live H W S STEPS                            oracle run side by side (small cases)

The CSV holds per-step loss_hip, loss_oracle, |diff|; the trailer the maximum, the first step (if any) past 1e-3, and the RMS of the
final image (contrast 1.1) against the oracle's 4x4 block means (full-resolution RMS: compare the dumped image with the oracle's,
`oracle/make_loss_curves.py --full-dir`).
"""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.weights import stress_visual_weights, visual_config
from oracle import reference_path as R


def seed_all(s):
    torch.manual_seed(s); np.random.seed(s)


def block_mean(img, k=4):
    c, h, w = img.shape
    return img[:, :h // k * k, :w // k * k].reshape(c, h // k, k, w // k, k).mean((2, 4))


def hip_engine(h, w, S, weights='synthetic', transform=None, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
        if weights == 'stress':
            model = aclip.CLIPModel('ViT-B/32', visual_config('ViT-B/32'), stress_visual_weights(visual_config('ViT-B/32'), 1), None, S)
    seed_all(0)
    p0 = R.fft_params_init([1, 3, h, w])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    eng = Engine(p0.cuda().contiguous(), h, w, model, S, [(target, -1.0)], sim='mix', transform=transform or transforms.normalize(), rng='reference', **kw)
    return eng, model, p0, target


def run_fixture(name, csv_out=None, img_out=None, steps=None, **kw):
    """-> (max |diff|, first step past 1e-3 or None, block-mean RMS, loss_hip array)"""
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'loss_curve_%s.npz' % name))
    want = fx['loss']
    meta = str(fx['meta'])
    h, w = 720, 1280
    S = int(meta.split(' cuts')[0].split(', ')[-1])
    steps = len(want) if steps is None else min(steps, len(want))
    fast = '-tf fast' in meta          # the per-cut augment draws interleave with the crop draws in the reference's order (utils.py:244-251)
    eng, _, _, _ = hip_engine(h, w, S, 'stress' if 'stress' in name else 'synthetic', transforms.transforms_fast if fast else None, **kw)
    seed_all(9)
    got = np.zeros(steps)
    for i in range(steps):
        if fast:
            from aphantasia_amd.utils import draw_crop_params
            table, augs = draw_crop_params(S, 224, h, w, 'uniform', 0.4, transforms.transforms_fast)
            got[i] = float(eng.step(table, augs))
        else:
            table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
            got[i] = float(eng.step(table))
    diff = np.abs(got - want[:steps])
    over = np.nonzero(diff > 1e-3)[0]
    rms = None
    if steps == len(want):
        with torch.no_grad():
            img = eng.synthesize(1.1).float().cpu()
        rms = float((block_mean(img) - torch.from_numpy(fx['img_blk'])).pow(2).mean().sqrt())
        if img_out:
            np.save(img_out, img.numpy().astype(np.float16))
    if csv_out:
        with open(csv_out, 'w') as f:
            f.write('# %s\n# HIP path: fp16 MFMA operands / fp32 accumulate, loss scale %g, %d skipped steps\n' % (meta, eng.loss_scale, int(eng.guard[0])))
            f.write('step,loss_hip,loss_oracle,abs_diff\n')
            for i in range(steps):
                f.write('%d,%.7f,%.7f,%.2e\n' % (i, got[i], want[i], diff[i]))
            f.write('# max |diff| %.2e at step %d ; first step past 1e-3: %s ; final image RMS vs the oracle (4x4 block means, contrast 1.1): %s\n'
                    % (diff.max(), int(diff.argmax()), int(over[0]) if len(over) else 'none', '%.5f' % rms if rms is not None else 'n/a'))
    return float(diff.max()), (int(over[0]) if len(over) else None), rms, got


if __name__ == '__main__':
    if sys.argv[1] == '--live':
        from oracle import clip_vit_ref
        h, w, S, steps = [int(v) for v in sys.argv[2:6]]
        eng, model, p0, target = hip_engine(h, w, S)
        cfg, wts = model.visual.cfg, model.visual.weights
        run = R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
        seed_all(9)
        print('step,loss_hip,loss_oracle,abs_diff')
        for i in range(steps):
            table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
            a, b = float(eng.step(table)), run.step(table)
            print('%d,%.7f,%.7f,%.2e' % (i, a, b, abs(a - b)))
    else:
        name = sys.argv[1]
        mx, first, rms, _ = run_fixture(name, sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
        print('%s: max |d loss| %.2e, first step past 1e-3: %s, final block-mean RMS %s' % (name, mx, first, rms))
