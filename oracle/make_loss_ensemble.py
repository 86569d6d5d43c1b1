"""TEST INFRASTRUCTURE (not the product).  An ENSEMBLE of free-running loss-vs-step trajectories of the fp32 torch-CPU oracle
(`oracle.reference_path.ReferenceRun` = the reference's train(i), clip_fft.py:235-295,308-310) on the stress weights, so that the
precision mode of the HIP path is decided -- and its 1e-3 tolerance asserted -- over many independent (weights, crops, shard size)
draws instead of one trajectory (one curve per mode cannot tell 2.6e-4 from 9.1e-4 apart from a coin flip).

    python oracle/make_loss_ensemble.py [--procs 2] [--only ws:cs:S ...]

Members (all 1280x720, ViT-B/32, `-tf none`, sim 'mix', Adam(lr .05, b1 0), 60 free-running steps, parameters from
`R.fft_params_init` after seed_all(0), target randn(512) seed 2 -- exactly the set-up of oracle/make_loss_curves.py's stress fixtures):
    weight seeds 1..8  x  cuts {32, 48, 95}  with crop seed 9          (24 members: `weights.stress_visual_weights(cfg, seed)`)
    weight seed 1      x  cuts {32, 48, 95}  with crop seeds 10, 11    ( 6 members: other crop draws on one weight set)
32 cuts = an 8-rank shard's small-M kernels, 48 cuts = the smallest batch on the full-batch (wave-specialised) GEMM kernels,
95 cuts = a 2-rank shard of the headline.

One tiny file per member, tests/golden/ensemble/stress_w<ws>_c<cs>_s<S>.npz: `loss` [60] f64, final-image channel mean / std at
contrast 1.1, meta.  Existing files are skipped (the run is resumable); tools/loss_ensemble.py is the GPU-side consumer.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphantasia_amd.weights import stress_visual_weights, visual_config   # noqa: E402  (weights only: no HIP)
from oracle import clip_vit_ref                                             # noqa: E402
from oracle import reference_path as R                                      # noqa: E402

H, W, STEPS = 720, 1280, 60
OUT = os.path.join(ROOT, 'tests', 'golden', 'ensemble')


def members():
    m = [(ws, 9, S) for S in (32, 48, 95) for ws in range(1, 9)]
    m += [(1, cs, S) for S in (32, 48, 95) for cs in (10, 11)]
    return m


def path_of(ws, cs, S):
    return os.path.join(OUT, 'stress_w%d_c%d_s%d.npz' % (ws, cs, S))


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def run(ws, cs, S, threads):
    out = path_of(ws, cs, S)
    if os.path.exists(out):
        return
    torch.set_num_threads(threads)
    cfg = visual_config('ViT-B/32')
    wts = stress_visual_weights(cfg, ws)
    seed_all(0)
    p0 = R.fft_params_init([1, 3, H, W])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    ref = R.ReferenceRun(H, W, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
    seed_all(cs)
    loss = np.zeros(STEPS)
    t0 = time.time()
    for i in range(STEPS):
        loss[i] = ref.step(R.draw_crop_table(S, 224, H, W, 'uniform', 0.4))
        if i % 10 == 0 or i == STEPS - 1:
            print('w%d c%d s%d step %d/%d loss %.6f  (%.1f s/step)' % (ws, cs, S, i, STEPS, loss[i], (time.time() - t0) / (i + 1)), flush=True)
    with torch.no_grad():
        img = ref.image(1.1)[0].float()
    meta = ('%dx%d ViT-B/32 stress weights (seed %d), %d cuts, crop seed %d, -tf none, sim mix, Adam(lr .05, b1 0), %d free-running steps; '
            'fp32 torch-CPU oracle (ReferenceRun), torch %s, %d threads' % (W, H, ws, S, cs, STEPS, torch.__version__, threads))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(out + '.tmp.npz', loss=loss, img_mean=img.mean((1, 2)).numpy(), img_std=img.std((1, 2)).numpy(), meta=np.array(meta))
    os.replace(out + '.tmp.npz', out)
    print('wrote', out, flush=True)


def worker(args):
    ws, cs, S, threads = args
    run(ws, cs, S, threads)


if __name__ == '__main__':
    argv = sys.argv[1:]
    procs = int(argv[argv.index('--procs') + 1]) if '--procs' in argv else 2
    if '--only' in argv:
        todo = [tuple(int(v) for v in a.split(':')) for a in argv[argv.index('--only') + 1:]]
    else:
        todo = members()
    todo = [m for m in todo if not os.path.exists(path_of(*m))]
    threads = max(1, (os.cpu_count() or 2) // procs)
    if procs == 1:
        for m in todo:
            run(*m, threads)
    else:
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(procs) as pool:
            pool.map(worker, [m + (threads,) for m in todo], chunksize=1)
