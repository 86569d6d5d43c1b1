"""TEST INFRASTRUCTURE (not the product).  Free-running loss-vs-step trajectories of the fp32 torch-CPU oracle
(`oracle.reference_path.ReferenceRun` = the reference's train(i), clip_fft.py:235-295,308-310) at BASELINE.json's
C2 workload, written as small fixtures under tests/golden/ so that the `-m gpu` tests and tools/loss_curve.py can
compare the HIP path's own free-running curve with them WITHOUT spending GPU-box minutes on ~8-18 s/step of host CPU.

    python oracle/make_loss_curves.py [name ...]        # default: every configuration below

Configurations (seeded synthetic ViT-B/32 weights, `-tf none`, sim 'mix', Adam(lr .05, b1 0), 1280x720):
    c2_s200      200 cuts, 50 steps          (BASELINE configs[1] at its real sample count)
    c2_s200_200  200 cuts, 200 steps         (BASELINE configs[1] verbatim)
    c2_s32       32 cuts, 200 steps          (BASELINE configs[1]'s step count)
    c2_s32_stress  32 cuts, 60 steps, `weights.stress_visual_weights` (LN gains 0.2-10, massive channels, peaky attention)
    c2_s48_stress  the same at 48 cuts (a 4-rank shard of the headline: large enough for the full-batch GEMM kernel and its split-precision QKV form)
    c2_s190_fast 190 cuts, 60 steps, `-tf fast` (transforms.py:165-170 through oracle/augment_ref.apply_fast, the draws interleaved
                 per cut in the reference's order, utils.py:244-251): the configuration bench.py's headline `value` is measured on
Both sides draw the crop tables with `R.draw_crop_table` after `seed_all(9)`; the parameters start from
`R.fft_params_init` after `seed_all(0)` -- exactly what tools/loss_curve.py and the tests do on the GPU side.

Each fixture holds: `loss` [steps] f64, the final image at contrast 1.1 as 4x4 block means `img_blk` [3,H/4,W/4] f32 and
per-channel mean/std, plus the meta string.  The full-resolution final image goes to `--full-dir` (not committed: 11 MB each)
for the pixel-RMS figure quoted in profiles/r03_loss_curve_*.csv.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphantasia_amd.weights import stress_visual_weights, synthetic_visual_weights, visual_config   # noqa: E402  (weights only: no HIP)
from oracle import augment_ref, clip_vit_ref                                                               # noqa: E402
from oracle import reference_path as R                                                             # noqa: E402

CONFIGS = {
    'c2_s200': dict(h=720, w=1280, S=200, steps=50, weights='synthetic'),
    'c2_s200_200': dict(h=720, w=1280, S=200, steps=200, weights='synthetic'),     # BASELINE configs[1] verbatim: samples=200, steps=200 (~45 min of CPU)
    'c2_s32': dict(h=720, w=1280, S=32, steps=200, weights='synthetic'),
    'c2_s32_stress': dict(h=720, w=1280, S=32, steps=60, weights='stress'),
    'c2_s48_stress': dict(h=720, w=1280, S=48, steps=60, weights='stress'),        # [r5] 48 cuts = 2400 token rows: the QKV launch is on the wave-specialised kernel (the headline's arithmetic)
    'c2_s190_fast': dict(h=720, w=1280, S=190, steps=60, weights='synthetic', tf='fast'),   # --samples 200 -> int(200 * .95) cuts (clip_fft.py:169)
}


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def weights_of(kind, name='ViT-B/32'):
    cfg = visual_config(name)
    return cfg, (stress_visual_weights(cfg, 1) if kind == 'stress' else synthetic_visual_weights(cfg, 1))


def block_mean(img, k=4):
    c, h, w = img.shape
    return img[:, :h // k * k, :w // k * k].reshape(c, h // k, k, w // k, k).mean((2, 4))


def run(name, full_dir=None):
    c = CONFIGS[name]
    h, w, S, steps = c['h'], c['w'], c['S'], c['steps']
    cfg, wts = weights_of(c['weights'])
    seed_all(0)
    p0 = R.fft_params_init([1, 3, h, w])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    ref = R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
    seed_all(9)
    loss = np.zeros(steps)
    t0 = time.time()
    for i in range(steps):
        if c.get('tf') == 'fast':
            augs = []
            table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4, per_cut_hook=lambda _c: augs.append(augment_ref.draw_fast_params(224)))
            loss[i] = ref.step(table, lambda k, cut: augment_ref.apply_fast(cut, augs[k], R.normalize))
        else:
            table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
            loss[i] = ref.step(table)
        if i % 5 == 0 or i == steps - 1:
            print('%s step %d/%d loss %.6f  (%.1f s/step)' % (name, i, steps, loss[i], (time.time() - t0) / (i + 1)), flush=True)
    with torch.no_grad():
        img = ref.image(1.1)[0].float()
    meta = ('%dx%d ViT-B/32 %s weights (seed 1), %d cuts, -tf %s, sim mix, Adam(lr .05, b1 0), %d free-running steps; '
            'fp32 torch-CPU oracle (ReferenceRun), torch %s, %d threads' % (w, h, c['weights'], S, c.get('tf', 'none'), steps, torch.__version__, torch.get_num_threads()))
    out = os.path.join(ROOT, 'tests', 'golden', 'loss_curve_%s.npz' % name)
    np.savez_compressed(out, loss=loss, img_blk=block_mean(img).numpy().astype(np.float32), img_mean=img.mean((1, 2)).numpy(),
                        img_std=img.std((1, 2)).numpy(), meta=np.array(meta))
    if full_dir:
        os.makedirs(full_dir, exist_ok=True)
        np.save(os.path.join(full_dir, 'oracle_final_%s.npy' % name), img.numpy().astype(np.float16))
    print('wrote', out, flush=True)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    full = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--full-dir=')), None)
    for n in (args or [k for k in CONFIGS if k != 'c2_s200_200']):
        run(n, full)
