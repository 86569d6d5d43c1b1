#!/usr/bin/env python
"""Continuous-mode text -> video on MI355X: the per-frame hot path of the reference's illustrip.py (illustrip.py:342-480
`process()`), same flag names and defaults for everything on that path.

    python illustrip.py -t "a forest in the fog" --size 1280-720 --steps 100 [--gen RGB|FFT] [-dm 2] [--ranks 8]

Per frame: warp the current picture by the frame's motion (scale / shift / angle / shear), re-create the parameters from it,
restart the optimiser, take --opt_step optimisation steps against the prompt(s), save the frame (aphantasia_amd/
illustrip_loop.py).  Built on the fused HIP engine; multi-GPU as clip_fft.py (cuts split over ranks, one RCCL all-reduce).

-d / --depth runs the depth warp (depth/depth.py:41-84 on HIP kernels); its estimator, Depth-Anything-V2, is loaded by
`transformers` from a LOCAL checkpoint directory (--depth_weights or APH_DEPTH_WEIGHTS; there is no network here) and is the one
part of the frame that runs on stock PyTorch.
Out of this path's scope: multi-line text files with topic interpolation, `latent_anima` motion curves (the motion here is the constant one of `--anima False`), --aest, LPIPS,
-tf custom / elastic, translation.  The flags exist and are refused with a message instead of being silently ignored.
"""
import argparse
import os
import shutil
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('-s', '--size', default='1280-720')
    p.add_argument('-t', '--in_txt', default=None)
    p.add_argument('-t2', '--in_txt2', default=None)
    p.add_argument('-t0', '--in_txt0', default=None)
    p.add_argument('--out_dir', default='_out')
    p.add_argument('--invert', action='store_true')
    p.add_argument('-v', '--verbose', dest='verbose', action='store_true')
    p.add_argument('-nv', '--no-verbose', dest='verbose', action='store_false')
    p.set_defaults(verbose=True)
    p.add_argument('--gen', default='RGB')
    p.add_argument('-m', '--model', default='ViT-B/32', choices=['ViT-B/16', 'ViT-B/32'])
    p.add_argument('--steps', default=300, type=int, help='frames per scene')
    p.add_argument('--samples', default=100, type=int)
    p.add_argument('-lr', '--lrate', default=0.1, type=float)
    p.add_argument('-dm', '--dualmod', default=None, type=int)
    p.add_argument('-ops', '--opt_step', default=1, type=int)
    p.add_argument('-sm', '--smooth', action='store_true')
    p.add_argument('--scale', default=0.012, type=float)
    p.add_argument('--shift', default=10., type=float)
    p.add_argument('--angle', default=0.8, type=float)
    p.add_argument('--shear', default=0.4, type=float)
    p.add_argument('-d', '--depth', default=0, type=float, help='Add depth with such strength, if > 0')
    p.add_argument('--depth_model', default='b', help='Depth Anything model: large, base or small')
    p.add_argument('--depth_dir', default=None, help='Directory to save depth, if not None')
    p.add_argument('--depth_weights', default=None, help='local Depth-Anything-V2 checkpoint directory (HF format); or APH_DEPTH_WEIGHTS')
    p.add_argument('-a', '--align', default='overscan', choices=['central', 'uniform', 'overscan', 'overmax'])
    p.add_argument('-tf', '--transform', default='fast', choices=['none', 'fast', 'custom', 'elastic'])
    p.add_argument('-opt', '--optimizer', default='adam_custom', choices=['adam', 'adam_custom', 'adamw', 'adamw_custom'])
    p.add_argument('--fixcontrast', action='store_true')
    p.add_argument('--contrast', default=1.2, type=float)
    p.add_argument('--colors', default=2.3, type=float)
    p.add_argument('-sh', '--sharp', default=0, type=float)
    p.add_argument('-mc', '--macro', default=0.3, type=float)
    p.add_argument('--aest', default=0., type=float)
    p.add_argument('-e', '--enforce', default=0, type=float)
    p.add_argument('-x', '--expand', default=0, type=float)
    p.add_argument('-n', '--noise', default=2., type=float)
    p.add_argument('--sim', default='mix')
    # additive
    p.add_argument('--clip-weights', dest='clip_weights', default=None)
    p.add_argument('--clip-weights2', dest='clip_weights2', default=None)
    p.add_argument('--seed', default=None, type=int)
    p.add_argument('--no_save', action='store_true')
    p.add_argument('--rng', default=None, choices=['bulk', 'reference'])
    p.add_argument('--ranks', default=1, type=int, help='GPUs of this node to shard the cuts over (launch with torchrun, or let this flag spawn the ranks)')
    p.add_argument('--graph-allreduce', action='store_true', help='with --ranks N: the step incl. its RCCL all-reduce as one hipGraph (opt-in; APH_MULTIRANK_GRAPH=1)')
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay (debugging)')
    a = p.parse_args(argv)
    a.size = [int(s) for s in a.size.split('-')][::-1]                    # illustrip.py:90-91
    if len(a.size) == 1: a.size = a.size * 2
    a.gen = a.gen.upper()
    a.invert = -1. if a.invert is True else 1.
    if a.gen == 'RGB':                                                     # illustrip.py:96-100
        a.smooth = False
        a.align = 'overscan'
    if a.model == 'ViT-B/16': a.sim = 'cossim'
    if a.dualmod is not None:                                              # illustrip.py:106-108
        a.model = 'ViT-B/32'
        a.sim = 'cossim'
    if a.rng is None:
        a.rng = 'reference' if a.seed is not None else 'bulk'
    return a


def derate_samples(a):
    """illustrip.py:152-160,177-179"""
    xmem = {'ViT-B/16': 0.25}
    s = a.samples
    if a.model in xmem: s = int(s * xmem[a.model])
    if a.dualmod is not None: s = int(s * 0.23)
    if a.enforce != 0: s = int(s * 0.5)
    if a.transform in ('elastic', 'custom', 'fast'): s = int(s * 0.95)
    return s


def _spawn_rank(local_rank, argv, world, port, run_id):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), APH_RUN_ID=run_id, HSA_ENABLE_IPC_MODE_LEGACY='0')
    main(argv)


def main(argv=None):
    a = get_args(argv)
    # multi-GPU as clip_fft.py: every rank holds the same picture, draws the same crop tables (same seed) and takes its share of the
    # cuts; one RCCL all-reduce of the parameter gradient per step.  The per-frame warp / re-parameterisation is replicated.
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if a.ranks > 1 and 'RANK' not in os.environ:
        from aphantasia_amd.comm import spawn_ranks
        args = list(sys.argv[1:] if argv is None else argv)
        if a.seed is None:
            args += ['--seed', str(int.from_bytes(os.urandom(3), 'little'))]
        port = 20000 + int.from_bytes(os.urandom(2), 'little') % 20000
        spawn_ranks(_spawn_rank, (args, a.ranks, port, 'i%d' % os.getpid()), a.ranks)
        return
    comm = None
    if world > 1:
        if a.seed is None:
            raise SystemExit(' multi-rank runs need --seed (every rank draws the same crop / augment tables and takes its share of the cuts)')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
        from aphantasia_amd import comm as acomm
        comm = acomm.create(rank, world)
        if rank != 0:
            a.verbose, a.no_save, a.depth_dir = False, True, None            # rank 0 reports and writes the frames / depth maps
    if a.aest != 0 or a.transform in ('custom', 'elastic'):
        raise SystemExit(' --aest / -tf custom|elastic are not part of this path')
    if a.in_txt is None and a.in_txt2 is None:
        raise SystemExit(' give a prompt with -t (a text file with several lines / topic interpolation is not part of this path)')
    for t in (a.in_txt, a.in_txt2, a.in_txt0):
        if t is not None and os.path.isfile(t):
            raise SystemExit(' text files (one prompt per line, interpolated) are not part of this path: pass the prompt itself')
    if a.seed is not None:
        torch.manual_seed(a.seed); np.random.seed(a.seed)
    import clip_fft
    from aphantasia_amd import clip as aclip, transforms
    from aphantasia_amd.engine import Engine
    from aphantasia_amd.illustrip_loop import FrameLoop
    from aphantasia_amd.utils import txt_clean
    with warnings.catch_warnings():
        if a.clip_weights is None:
            print(' !! no --clip-weights given: using seeded SYNTHETIC CLIP weights (timing / plumbing only)')
            warnings.simplefilter('ignore')
        model = aclip.load(a.model, weights=a.clip_weights)[0]
        model2 = aclip.load('ViT-B/16', weights=a.clip_weights2)[0] if a.dualmod is not None else None
    S = derate_samples(a)
    clip_fft.check_samples(S)
    h, w = a.size

    def targets_for(m):
        out = []
        for txt, sign in ((a.in_txt, -a.invert), (a.in_txt2, -1.0), (a.in_txt0, 1.0)):          # illustrip.py:442-450
            if txt is None: continue
            for sub in txt.split('|'):
                wt = 1.
                if ':' in sub: sub, wt = sub.split(':')[0], float(sub.split(':')[1])
                out.append((aclip.text_embedding(m, sub), sign * wt))
        return out
    trf = transforms.transforms_fast if a.transform == 'fast' else transforms.normalize()
    if a.gen == 'RGB':                                                                          # illustrip.py:270-272: pixel_image([1,3,*size], resume) -> randn * sd, sd = 1 (image.py:98-101)
        leaf = torch.randn(1, 3, h, w).cuda().contiguous()
        pk = dict(param_kind='pixel', rgb_priors=True, fixcontrast=a.fixcontrast, decay=1.0)
    else:
        leaf = (0.01 * torch.randn(1, 3, h, w // 2 + 1, 2)).cuda().contiguous()
        pk = dict(param_kind='fft', decay=1.0)                                                  # fft_image default decay_power (illustrip.py:409)
    kw = dict(sim=a.sim, colors=a.colors, lr=a.lrate, optimizer=a.optimizer, align=a.align, macro=a.macro, transform=trf, sharp=a.sharp,
              expand=a.expand, enforce=a.enforce, rng=a.rng, rank=rank, world=world, comm=comm, graph_allreduce=a.graph_allreduce or None, use_graph=not a.no_graph, **pk)
    eng = Engine(leaf, h, w, model, S, targets_for(model), **kw)
    eng2 = Engine(leaf, h, w, model2, S, targets_for(model2), state=eng.state(), **kw) if model2 is not None else None
    depth_fn = None
    if a.depth > 0:
        from aphantasia_amd.depthwarp import InferDepthAny
        depth_fn = InferDepthAny(a.depth_model, path=a.depth_weights)          # raises without a local checkpoint
        if a.depth_dir is not None:
            os.makedirs(a.depth_dir, exist_ok=True)
    loop = FrameLoop(eng, gen=a.gen, opt_step=a.opt_step, smooth=a.smooth, engine2=eng2, dualmod=a.dualmod, depth=a.depth, depth_fn=depth_fn,
                     colors=a.colors, depth_dir=a.depth_dir)
    name = txt_clean(a.in_txt or a.in_txt2).lower()[:40] + '-%s' % a.gen
    tempdir = os.path.join(a.out_dir, name)
    os.makedirs(tempdir, exist_ok=True)
    writer = None if a.no_save else clip_fft.FrameWriter(h, w)
    t0 = time.time()
    for num in range(a.steps):
        img = loop.frame(1 + a.scale, [0, a.shift], a.angle, a.shear, contrast=None if writer is None else a.contrast,     # illustrip.py:381-384 (anima off)
                         noise=a.noise if a.gen == 'FFT' else 0.0, consume_noise_draw=(a.gen != 'FFT' and a.noise > 0 and a.rng == 'reference'))
        if writer is not None:
            writer.put(img.reshape(3, h, w), os.path.join(tempdir, '%06d.jpg' % num), 1.0)
        if (a.verbose or world > 1) and (num % 10 == 9 or num == a.steps - 1):
            gl = eng.global_loss()             # (a collective when world > 1: every rank takes part, rank 0 prints)
            if a.verbose:
                print(' frame %d/%d  loss %.4f  %.1f frames/s' % (num + 1, a.steps, gl, (num + 1) / (time.time() - t0)), flush=True)
    torch.cuda.synchronize()
    if writer is not None:
        writer.close()
        if shutil.which('ffmpeg'):
            os.system('ffmpeg -v warning -y -i %s/\\%%06d.jpg "%s.mp4"' % (tempdir, os.path.join(a.out_dir, name)))
    if rank == 0:
        print(' done: %d frames in %.1fs (%.1f frames/s)%s' % (a.steps, time.time() - t0, a.steps / (time.time() - t0), ' on %d ranks' % world if world > 1 else ''))


if __name__ == '__main__':
    main()
