#!/usr/bin/env python
"""Text/image -> image optimisation on MI355X: drop-in for the reference's clip_fft.py.

Same command line (every flag of the reference's get_args, clip_fft.py:37-77, with the same defaults,
post-parse overrides and sample-count derating arithmetic), same output naming and `.pt` snapshot
format.  Additive flags: --clip-weights PATH (OpenAI CLIP checkpoint; without it seeded synthetic
weights are used and the pictures are meaningless), --seed N (seeds torch + numpy; the reference is
unseeded), --no_save (skip the per-step JPEG, for timing).

The whole loss -- text / style / subtract prompts, a reference image, any --sim, --sharp, --enforce, --expand, --noise and
--aest (with --aest-weights: the LAION linear head cannot be downloaded here) -- runs through the fused HIP engine
(aphantasia_amd/engine.py).  --sync (LPIPS: a separate VGG network) is not part of this path.
"""
import argparse
import os
import shutil
import sys
import threading
import queue
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

clip_models = ['ViT-B/16', 'ViT-B/32', 'RN101', 'RN50x16', 'RN50x4', 'RN50']


def get_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('-t',  '--in_txt',  default=None, help='input text')
    parser.add_argument('-t2', '--in_txt2', default=None, help='input text - style')
    parser.add_argument('-t0', '--in_txt0', default=None, help='input text to subtract')
    parser.add_argument('-i',  '--in_img',  default=None, help='input image')
    parser.add_argument('-wi', '--weight_img', default=0.5, type=float, help='weight for images')
    parser.add_argument(       '--out_dir', default='_out')
    parser.add_argument('-s',  '--size',    default='1280-720', help='Output resolution')
    parser.add_argument('-r',  '--resume',  default=None, help='Path to saved FFT snapshots, to resume from')
    parser.add_argument('-ops', '--opt_step', default=1, type=int, help='How many optimizing steps per save step')
    parser.add_argument('-tr', '--translate', action='store_true', help='Translate text with Google Translate')
    parser.add_argument(       '--save_pt', action='store_true', help='Save FFT snapshots for further use')
    parser.add_argument('-v',  '--verbose',    dest='verbose', action='store_true')
    parser.add_argument('-nv', '--no-verbose', dest='verbose', action='store_false')
    parser.set_defaults(verbose=True)
    # training
    parser.add_argument('-m',  '--model',   default='ViT-B/32', choices=clip_models, help='Select CLIP model to use')
    parser.add_argument(       '--steps',   default=200, type=int, help='Total iterations')
    parser.add_argument(       '--samples', default=200, type=int, help='Samples to evaluate')
    parser.add_argument('-lr', '--lrate',   default=0.05, type=float, help='Learning rate')
    parser.add_argument('-p',  '--prog',    action='store_true', help='Enable progressive lrate growth (up to double a.lrate)')
    parser.add_argument('-dm', '--dualmod', default=None, type=int, help='Every this step use another CLIP ViT model')
    # wavelet
    parser.add_argument(       '--dwt',     action='store_true', help='Use DWT instead of FFT')
    parser.add_argument('-w',  '--wave',    default='coif2', help='wavelets: db[1..], coif[1..], haar, dmey')
    # tweaks
    parser.add_argument('-a',  '--align',   default='uniform', choices=['central', 'uniform', 'overscan', 'overmax'], help='Sampling distribution')
    parser.add_argument('-tf', '--transform', default='fast', choices=['none', 'fast', 'custom', 'elastic'], help='augmenting transforms')
    parser.add_argument('-opt', '--optimizer', default='adam_custom', choices=['adam', 'adamw', 'adam_custom', 'adamw_custom'], help='Optimizer')
    parser.add_argument(       '--contrast', default=1.1, type=float)
    parser.add_argument(       '--colors',  default=1.8, type=float)
    parser.add_argument(       '--decay',   default=1.5, type=float)
    parser.add_argument('-sh', '--sharp',   default=0., type=float)
    parser.add_argument('-mm', '--macro',   default=0.4, type=float, help='Endorse macro forms 0..1 ')
    parser.add_argument(       '--aest',    default=0., type=float, help='Enhance aesthetics')
    parser.add_argument('-e',  '--enforce', default=0, type=float, help='Enforce details (by boosting similarity between two parallel samples)')
    parser.add_argument('-x',  '--expand',  default=0, type=float, help='Boosts diversity (by enforcing difference between prev/next samples)')
    parser.add_argument('-n',  '--noise',   default=0, type=float, help='Add noise to suppress accumulation')
    parser.add_argument('-c',  '--sync',    default=0, type=float, help='Sync output to input image')
    parser.add_argument(       '--invert',  action='store_true', help='Invert criteria')
    parser.add_argument(       '--sim',     default='mix', help='Similarity function (dot/angular/spherical/mixed; None = cossim)')
    # additive (not in the reference)
    parser.add_argument(       '--clip-weights', dest='clip_weights', default=None, help='OpenAI CLIP checkpoint (ViT-B-32.pt); second model: --clip-weights2')
    parser.add_argument(       '--clip-weights2', dest='clip_weights2', default=None, help='checkpoint of the --dualmod model (ViT-B-16.pt)')
    parser.add_argument(       '--seed',    default=None, type=int, help='seed torch/numpy RNG (reference: unseeded)')
    parser.add_argument(       '--no_save', action='store_true', help='do not write the per-step JPEG frames')
    parser.add_argument(       '--precise', action='store_true', help='opt-in split-precision ViT forward (patch-embedding and QKV GEMMs on hi + lo f16 activation pairs): ~5 %% slower, lower single-step gradient error; '
                               'on the stress-weight loss-curve ensemble not significantly closer to the fp32 CPU reference than the default')
    parser.add_argument(       '--fast-f16', action='store_true', help='(the default since round 6; accepted for old command lines) f16 operands on every ViT GEMM, as the reference runs CLIP on a GPU')
    parser.add_argument(       '--aest-weights', dest='aest_weights', default=None, help="state dict of the LAION aesthetic head (sa_0_4_vit_b_32_linear.pth: "
                               "{'weight': [1,512], 'bias': [1]}); upstream downloads it (utils.py:402-413), there is no network here")
    parser.add_argument(       '--aest-weights2', dest='aest_weights2', default=None, help='the head of the --dualmod model (sa_0_4_vit_b_16_linear.pth)')
    parser.add_argument(       '--rng',     default=None, choices=['bulk', 'reference'], help="host random draws: 'reference' = the reference's exact draw "
                               "order on torch's / numpy's global generators (a seeded run reproduces the reference's crop and augment tables); "
                               "'bulk' = vectorised draws from a numpy Generator (same distributions, a different stream, ~10x less host time). "
                               "Default: reference when --seed is given, else bulk")
    parser.add_argument(       '--ranks',   default=1, type=int, help='GPUs of this node to shard the cuts over (launch with torchrun, or let this flag spawn the ranks)')
    parser.add_argument(       '--graph-allreduce', action='store_true', help='with --ranks N: capture the whole step INCLUDING its RCCL all-reduce into one hipGraph '
                               '(opt-in: multi-rank steps launch eagerly by default; same as APH_MULTIRANK_GRAPH=1)')
    parser.add_argument(       '--no-graph', action='store_true', help='eager launches instead of hipGraph replay (debugging)')
    a = parser.parse_args(argv)

    if a.size is not None: a.size = [int(s) for s in a.size.split('-')][::-1]        # clip_fft.py:80
    if len(a.size) == 1: a.size = a.size * 2
    if (a.in_img is not None and a.sync != 0) or a.resume is not None: a.align = 'overscan'
    if a.translate is True:
        print('\n Install googletrans module to enable translation!'); exit()
    if a.dualmod is not None:                                                         # clip_fft.py:86-88
        a.model = 'ViT-B/32'
        a.sim = 'cossim'
    if a.rng is None:
        a.rng = 'reference' if a.seed is not None else 'bulk'
    return a


def derate_samples(a):
    """The reference's sample-count arithmetic, in its order (clip_fft.py:125-127,134,157-169,187,199)."""
    xmem = {'ViT-B/16': 0.25, 'RN50': 0.5, 'RN50x4': 0.16, 'RN50x16': 0.06, 'RN101': 0.33}
    s = a.samples
    if a.model in xmem:
        s = int(s * xmem[a.model])
    if a.dualmod is not None:
        s = int(s * 0.23)
    if a.enforce != 0:
        s = int(s * 0.5)
    if a.sync > 0:
        s = int(s * 0.5)
    if a.transform in ('elastic', 'custom', 'fast'):
        s = int(s * 0.95)
    if a.in_txt2 is not None:
        s = int(s * 0.75)
    if a.in_txt0 is not None:
        s = int(s * 0.75)
    return s


def check_samples(n):
    """upstream, `--samples 1` with the default `-tf fast` derates to int(1 * 0.95) = 0 cuts and dies in torch.cat([])
    (clip_fft.py:169, utils.py:253); say so instead"""
    if n < 1:
        raise SystemExit(' --samples derates to %d cuts for this model / transform (clip_fft.py:125-169): nothing to optimise; '
                         'raise --samples or use -tf none' % n)


class FrameWriter:
    """Background JPEG writers: the reference converts and encodes every frame synchronously inside the hot loop
    (clip_fft.py:297-306, utils.py:94-100).  Here the loop only enqueues a device-side uint8 conversion and an async
    copy into a pinned ring; a small thread pool waits for the copy event and encodes (PIL releases the GIL)."""
    THREADS = 4        # (8 threads measured the same with-save rate: 153.6-154.0 vs 153.2 steps/s -- the encoders are not the limit)
    RING = 16          # slots of (device uint8 buffer, pinned host buffer); a slot is reused only after its writer released it (below)

    def __init__(self, h, w, switch_interval=None):
        # switch_interval: optionally lower the interpreter's GIL switch interval (default 5 ms) for the encoder threads' sake.  Measured in
        # round 6 (tools/exp/save_ab.py, profiles/r06_save_ab.txt): 158.7 / 158.6 / 158.7 steps/s at 5 / 0.5 / 0.1 ms against 163.8 without
        # saving -- the 3 % the per-step frame costs is device work (the contrast-1.1 synthesis, the uint8 conversion, the copy) and the
        # loop's own enqueue calls, not GIL hand-over; left at the interpreter's setting.
        if switch_interval is not None and sys.getswitchinterval() > switch_interval:
            sys.setswitchinterval(switch_interval)
        self.q = queue.Queue(maxsize=8)
        self.h, self.w = h, w
        self.bufs = [torch.empty(h, w, 3, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        # device-side uint8 ring + a copy stream of its own: the conversion is ONE launch on the step's stream (aph_rgb_to_u8), the
        # 2.7 MB device->host copy waits for it on the side stream, so the next step's launches do not queue behind the PCIe transfer
        self.dev = None
        self.copy_stream = None
        self.n = 0
        # slot discipline: `free[k]` is released by the worker AFTER imsave (a straggling encoder keeps its pinned buffer), and the copy
        # event of the slot's previous use is waited for on the step's stream before aph_rgb_to_u8 rewrites the device buffer
        self.free = [threading.Semaphore(1) for _ in range(self.RING)]
        self.copy_ev = [None] * self.RING
        self.ts = [threading.Thread(target=self._run, daemon=True) for _ in range(self.THREADS)]
        for t in self.ts: t.start()

    def _run(self):
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                self.q.task_done()
                return
            buf, ev, fname, k = item
            try:
                ev.synchronize()
                Image.fromarray(buf.numpy()).save(fname, quality=95)
            finally:
                self.free[k].release()
                self.q.task_done()

    def put(self, img, fname, gamma=1.0):
        """img: device float32 [3,H,W] in [0,1]; same arithmetic as utils.checkout: clip(img ** gamma * 255, 0, 255).astype(uint8)"""
        from aphantasia_amd import _ffi, ops
        if self.dev is None:
            self.dev = [torch.empty(self.h, self.w, 3, dtype=torch.uint8, device=img.device) for _ in range(self.RING)]
            self.copy_stream = torch.cuda.Stream(device=img.device)
        k = self.n % self.RING
        self.n += 1
        buf, dbuf = self.bufs[k], self.dev[k]
        img = img.contiguous()
        self.free[k].acquire()                                        # the encoder that used this slot RING frames ago has written its file
        if self.copy_ev[k] is not None:
            torch.cuda.current_stream(img.device).wait_event(self.copy_ev[k])      # ... and its device->host copy has read dbuf
        _ffi.lib().call('aph_rgb_to_u8', ops.ptr(img), self.h, self.w, float(gamma), ops.ptr(dbuf), ops._stream(img))
        done = torch.cuda.Event(); done.record()                      # on the step's stream: the conversion has read `img`
        self.copy_stream.wait_event(done)
        with torch.cuda.stream(self.copy_stream):
            buf.copy_(dbuf, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        self.copy_ev[k] = ev
        self.q.put((buf, ev, fname, k))

    def drain(self):
        """block until every frame handed to put() is on disk"""
        self.q.join()

    def close(self):
        for _ in self.ts: self.q.put(None)
        for t in self.ts: t.join()


def _spawn_rank(local_rank, argv, world, port, run_id):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), APH_RUN_ID=run_id, HSA_ENABLE_IPC_MODE_LEGACY='0')
    main(argv)


def main(argv=None):
    a = get_args(argv)
    # ---- multi-GPU (the reference is single-GPU): the cuts are split over the ranks of one node, one process
    # per GPU, ONE RCCL all-reduce of the parameter gradient per step (aph_allreduce_f32).  Either launched by torchrun (RANK /
    # WORLD_SIZE / LOCAL_RANK in the environment) or spawned from here with --ranks N.
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if a.ranks > 1 and 'RANK' not in os.environ:
        from aphantasia_amd.comm import spawn_ranks
        if a.seed is None:
            argv = list(sys.argv[1:] if argv is None else argv) + ['--seed', str(int.from_bytes(os.urandom(3), 'little'))]    # every rank must draw the same crop tables
        port = 20000 + int.from_bytes(os.urandom(2), 'little') % 20000
        spawn_ranks(_spawn_rank, (list(sys.argv[1:] if argv is None else argv), a.ranks, port, 'r%d' % os.getpid()), a.ranks)
        return
    comm = None
    if world > 1:
        if a.seed is None:
            raise SystemExit(' multi-rank runs need --seed (every rank draws the same crop / augment tables and takes its share of the cuts)')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
        from aphantasia_amd import comm as acomm
        comm = acomm.create(rank, world)
        if rank != 0:
            a.verbose, a.no_save, a.save_pt = False, True, False          # rank 0 reports and writes the frames
    if a.seed is not None:
        torch.manual_seed(a.seed)
        np.random.seed(a.seed)
    if not a.model.startswith('ViT'):
        raise SystemExit(' the MI355X path covers the ViT CLIP models (ViT-B/32, ViT-B/16); got %s' % a.model)
    if a.transform in ('custom', 'elastic'):
        raise SystemExit(' -tf %s (kornia-based) is not provided; use fast or none' % a.transform)
    from aphantasia_amd import clip as aclip, transforms
    from aphantasia_amd.image import to_valid_rgb, fft_image, dwt_image
    from aphantasia_amd.utils import slice_imgs, sim_func, basename, img_list, img_read, txt_clean
    from aphantasia_amd.engine import Engine

    shape = [1, 3, *a.size]
    if a.dwt is True:
        params, image_f, sz = dwt_image(shape, a.wave, 0.3, a.colors, a.resume)
    else:
        params, image_f, sz = fft_image(shape, 0.07, a.decay, a.resume)
    if sz is not None: a.size = sz
    rgb_f = to_valid_rgb(image_f, colors=a.colors)

    if a.prog is True:
        lr1 = a.lrate * 2
        lr0 = lr1 * 0.01
    else:
        lr0 = a.lrate
    sign = 1. if a.invert is True else -1.

    with warnings.catch_warnings():
        if a.clip_weights is None:
            print(' !! no --clip-weights given: using seeded SYNTHETIC CLIP weights (timing / plumbing only)')
            warnings.simplefilter('ignore')
        model_clip, _ = aclip.load(a.model, weights=a.clip_weights)
        a.modsize = model_clip.visual.input_resolution
        if a.verbose is True: print(' using model', a.model)
        a.samples = derate_samples(a)
        check_samples(a.samples)
        if a.dualmod is not None:
            model_clip2, _ = aclip.load('ViT-B/16', weights=a.clip_weights2)
            dualmod_nums = list(range(a.steps))[a.dualmod::a.dualmod]
            print(' dual model every %d step' % a.dualmod)

    def enc_text(txt, model=model_clip):                                              # clip_fft.py:143-154
        embs = []
        for subtxt in txt.split('|'):
            if ':' in subtxt:
                [subtxt, wt] = subtxt.split(':')
                wt = float(wt)
            else: wt = 1.
            embs.append([aclip.text_embedding(model, subtxt), wt])
        return embs

    trform_f = transforms.transforms_fast if 'fast' in a.transform else transforms.normalize()
    out_name = []
    targets, targets2 = [], []          # (embedding, coef) for model 1 / model 2
    if a.in_txt is not None:
        if a.verbose is True: print(' topic text: ', a.in_txt)
        targets += [(e, sign * w) for e, w in enc_text(a.in_txt)]
        out_name.append(txt_clean(a.in_txt).lower()[:40])
        if a.dualmod is not None: targets2 += [(e, sign * w) for e, w in enc_text(a.in_txt, model_clip2)]
    if a.in_txt2 is not None:
        if a.verbose is True: print(' style text:', a.in_txt2)
        targets += [(e, sign * w) for e, w in enc_text(a.in_txt2)]
        out_name.append(txt_clean(a.in_txt2).lower()[:40])
        if a.dualmod is not None: targets2 += [(e, sign * w) for e, w in enc_text(a.in_txt2, model_clip2)]
    if a.in_txt0 is not None:
        if a.verbose is True: print(' subtract text:', a.in_txt0)
        targets += [(e, -sign * w) for e, w in enc_text(a.in_txt0)]
        out_name.append('off-' + txt_clean(a.in_txt0).lower()[:40])
        if a.dualmod is not None: targets2 += [(e, -sign * w) for e, w in enc_text(a.in_txt0, model_clip2)]
    if a.in_img is not None and os.path.isfile(a.in_img):
        if a.verbose is True: print(' ref image:', basename(a.in_img))
        img_in = torch.from_numpy(img_read(a.in_img) / 255.).unsqueeze(0).permute(0, 3, 1, 2).cuda().float()[:, :3]
        with torch.no_grad():
            in_sliced = slice_imgs([img_in], a.samples, a.modsize, transforms.normalize(), a.align,
                                   patch=model_clip.visual.patch_size)[0]
            targets.append((model_clip.encode_image(in_sliced).detach().clone(), sign * a.weight_img))   # per-cut pairs, clip_fft.py:216,267
            if a.dualmod is not None:
                targets2.append((model_clip2.encode_image(in_sliced).detach().clone(), sign * a.weight_img))
        out_name.append(basename(a.in_img).replace(' ', '_'))
    assert targets, ' Loss not defined, check the inputs'                              # clip_fft.py:286

    if a.verbose is True: print(' samples:', a.samples)
    out_name = '-'.join(out_name)
    out_name += '-%s' % a.model.replace('/', '').replace('-', '') if a.dualmod is None else '-dm%d' % a.dualmod
    tempdir = os.path.join(a.out_dir, out_name)
    os.makedirs(tempdir, exist_ok=True)

    if a.sync != 0:
        raise SystemExit(' --sync (LPIPS: a separate VGG network, clip_fft.py:268-270) is not part of the MI355X path')

    def load_aest(path):                                                              # utils.py:402-413 aesthetic_model()
        if a.aest == 0:
            return None
        if path is None or not os.path.isfile(path):
            raise SystemExit(' --aest needs the LAION linear head: pass its state dict with --aest-weights (upstream downloads '
                             'sa_0_4_<model>_linear.pth from github.com/LAION-AI/aesthetic-predictor; there is no network here)')
        sd = torch.load(path, map_location='cpu')
        return (sd['weight'].float(), float(sd['bias'].reshape(-1)[0]), a.aest)
    aest1 = load_aest(a.aest_weights)
    aest2 = load_aest(a.aest_weights2) if a.dualmod is not None else None
    h, w = a.size
    if a.dwt is True:
        pk = dict(param_kind='dwt', dwt=image_f.synth)
        leaf = image_f.flat
    else:
        pk = dict(param_kind='fft')
        leaf = params[0]
    eng = Engine(leaf, h, w, model_clip, a.samples, targets, sim=a.sim, colors=a.colors, decay=a.decay, lr=lr0,
                 optimizer=a.optimizer, align=a.align, macro=a.macro, transform=trform_f, sharp=a.sharp, expand=a.expand, enforce=a.enforce, rng=a.rng,
                 rank=rank, world=world, comm=comm, aest=aest1, precise=a.precise, graph_allreduce=a.graph_allreduce or None, use_graph=not a.no_graph, **pk)
    h, w = eng.h, eng.w
    eng2 = None
    if a.dualmod is not None:
        eng2 = Engine(leaf, h, w, model_clip2, a.samples, targets2, sim=a.sim, colors=a.colors, decay=a.decay, lr=lr0,
                      optimizer=a.optimizer, align=a.align, macro=a.macro, transform=trform_f, state=eng.state(), sharp=a.sharp, expand=a.expand, enforce=a.enforce, rng=a.rng,
                      rank=rank, world=world, comm=comm, aest=aest2, precise=a.precise, graph_allreduce=a.graph_allreduce or None, use_graph=not a.no_graph, **pk)

    writer = None if a.no_save else FrameWriter(h, w)
    # empirical tone mapping of the saved frames (clip_fft.py:300-303): **1.3 with --sync, **(1 + sharp/2) with --sharp
    gamma = 1.3 if (a.sync > 0 and a.in_img is not None) else (1 + a.sharp / 2. if a.sharp != 0 else 1.0)
    t0 = time.time()
    for i in range(a.steps):
        e = eng2 if (eng2 is not None and i in dualmod_nums) else eng
        lr_cur = lr0 + (i / a.steps) * (lr1 - lr0) if a.prog is True else lr0      # clip_fft.py:288-291
        shift = None
        if a.noise > 0 and a.dwt is not True:
            shift = (a.noise * torch.rand(1, 1, h, w // 2 + 1, 1)).reshape(h, w // 2 + 1).cuda().contiguous()
        e.step(lr=lr_cur, shift=shift)
        if a.expand > 0:                                                                # clip_fft.py:276-280: prev_enc = out_enc.detach()
            for other in (eng, eng2):
                if other is not None: other.set_prev_enc(e.enc)
        if i % a.opt_step == 0 and writer is not None:
            img = e.synthesize(a.contrast)                                              # clip_fft.py:298-299
            writer.put(img.reshape(3, h, w), os.path.join(tempdir, '%04d.jpg' % (i // a.opt_step)), gamma)
        if (a.verbose or world > 1) and (i % 10 == 9 or i == a.steps - 1):
            gl = e.global_loss()               # (a collective when world > 1: every rank takes part, rank 0 prints)
            if a.verbose:
                print(' step %d/%d  loss %.4f  %.1f steps/s' % (i + 1, a.steps, gl, (i + 1) / (time.time() - t0)), flush=True)
    torch.cuda.synchronize()
    if writer is not None:
        writer.close()
        if shutil.which('ffmpeg'):
            os.system('ffmpeg -v warning -y -i %s/\\%%04d.jpg "%s.mp4"' % (tempdir, os.path.join(a.out_dir, out_name)))
        frames = img_list(tempdir)
        if frames:
            shutil.copy(frames[-1], os.path.join(a.out_dir, '%s-%d.jpg' % (out_name, a.steps)))
    if a.save_pt is True:
        torch.save([p.detach().cpu() for p in params], '%s.pt' % os.path.join(a.out_dir, out_name))   # clip_fft.py:315 (list of tensors)
    if rank == 0:
        print(' done: %d steps in %.1fs (%.1f steps/s)%s' % (a.steps, time.time() - t0, a.steps / (time.time() - t0), ' on %d ranks' % world if world > 1 else ''))


if __name__ == '__main__':
    main()
