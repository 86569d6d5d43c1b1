"""Randomised whole-step parity on the GPU: random frame sizes (any factorisation), cut counts, similarity types, optimisers, --align
modes, parameterisers, -tf none / fast and the optional loss terms -- two free-running steps of the fused engine (ViT-B/32, synthetic
weights) against the fp32 CPU oracle.  python tools/gpu_engine_fuzz.py [seed] [cases]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.utils import draw_crop_params
from oracle import reference_path as R, clip_vit_ref, augment_ref
transforms._EXACT_ZERO_ROT = True
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=16)
cfg, wts = model.visual.cfg, model.visual.weights
enc = lambda x: clip_vit_ref.encode_image(wts, x, cfg)


def seed_all(s):
    torch.manual_seed(s); np.random.seed(s)


bad = []
t0 = time.time()
for it in range(cases):
    H = int(rng.integers(225, 420)); W = int(rng.integers(225, 520)); S = int(rng.integers(1, 7))
    sim = str(rng.choice(['mix', 'cossim', 'ang'])); opt = str(rng.choice(['adam', 'adam_custom', 'adamw', 'adamw_custom']))
    align = str(rng.choice(['uniform', 'overscan', 'central', 'overmax'])); kind = str(rng.choice(['fft', 'pixel']))
    fast = bool(rng.integers(0, 2))
    sharp = float(rng.choice([0, 0, 0.3])); expand = float(rng.choice([0, 0, 0.5])) if sim != 'ang' else 0.0
    enforce = float(rng.choice([0, 0, 0.1])) if sim != 'ang' else 0.0
    case = dict(H=H, W=W, S=S, sim=sim, opt=opt, align=align, kind=kind, fast=fast, sharp=sharp, expand=expand, enforce=enforce)
    try:
        seed_all(it + 1000 * seed)
        p0 = R.fft_params_init([1, 3, H, W]).contiguous() if kind == 'fft' else torch.randn(1, 3, H, W) * 0.5
        tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
        trf = transforms.transforms_fast if fast else transforms.normalize()
        kw = dict(sim=sim, optimizer=opt, align=align, macro=0.4, sharp=sharp, expand=expand, enforce=enforce, transform=trf, rng='reference')
        okw = dict(sim=sim, optimizer=opt, align=align, sharp=sharp, expand=expand, enforce=enforce)
        if kind == 'pixel':
            kw.update(param_kind='pixel'); okw.update(param_kind='pixel')
        eng = Engine(p0.clone().cuda().contiguous(), H, W, model, S, [(tgt, -1.0)], **kw)
        run = R.ReferenceRun(H, W, enc, [(tgt, 1.0)], params=p0, **okw)
        per = lambda augs: None if augs is None else (lambda c, cut: augment_ref.apply_fast(cut, augs[c], R.normalize))
        for st in range(2):
            seed_all(100 + st)
            tb, augs = draw_crop_params(S, 224, H, W, align, 0.4, trf)
            tb2 = augs2 = None
            if enforce != 0:
                tb2, augs2 = draw_crop_params(S, 224, H, W, align, 0.4, trf)
            want = run.step(tb, per(augs), tb2, per(augs2))
            got = float(eng.step(tb, augs, tables2=None if tb2 is None else (tb2, augs2)))
            if expand > 0:
                eng.set_prev_enc()
            assert abs(got - want) < 2e-3, (st, got, want)
        assert int(eng.guard[0]) == 0, 'skipped step'
        print(it, case, 'ok %.0fs' % (time.time() - t0), flush=True)
    except AssertionError as e:
        bad.append((case, 'assert', str(e)[:120])); print(it, case, 'BAD', e, flush=True)
    except Exception as e:
        bad.append((case, type(e).__name__, str(e)[:200])); print(it, case, 'BAD', type(e).__name__, e, flush=True)
print('%d bad of %d in %.0f s' % (len(bad), cases, time.time() - t0))
