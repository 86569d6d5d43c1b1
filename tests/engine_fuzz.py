"""TEST INFRASTRUCTURE.  Randomised whole-step parity of the fused engine against the fp32 CPU oracle: random frame sizes (any
factorisation), cut counts, similarity types, optimisers, --align modes, parameterisers (FFT / pixel / DWT), -tf none / fast and the
optional loss terms -- two free-running steps per case (ViT-B/32, synthetic weights).

Used by `tests/test_gpu_parity_configs.py::test_engine_fuzz_seed` (fixed seeds, in `pytest -m gpu`: the driver runs them) and by
`tools/gpu_engine_fuzz.py` (any seed / case count by hand).  A case is a pure function of (seed, index)."""
import time
import warnings

import numpy as np
import torch

from aphantasia_amd import transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.utils import draw_crop_params
from oracle import augment_ref, clip_vit_ref
from oracle import reference_path as R

WAVES = ['db3', 'coif2', 'haar']


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def load_model(max_batch=16):
    from aphantasia_amd import clip as aclip
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model, _ = aclip.load('ViT-B/32', seed=1, max_batch=max_batch)
    return model


def draw_case(rng, force=None):
    """one random configuration; `force` pins some of its fields (e.g. dict(fast=True) or dict(kind='dwt'))"""
    H = int(rng.integers(225, 420)); W = int(rng.integers(225, 520)); S = int(rng.integers(1, 7))
    sim = str(rng.choice(['mix', 'cossim', 'ang'])); opt = str(rng.choice(['adam', 'adam_custom', 'adamw', 'adamw_custom']))
    align = str(rng.choice(['uniform', 'overscan', 'central', 'overmax'])); kind = str(rng.choice(['fft', 'pixel', 'dwt']))
    fast = bool(rng.integers(0, 2))
    sharp = float(rng.choice([0, 0, 0.3])); expand = float(rng.choice([0, 0, 0.5])) if sim != 'ang' else 0.0
    enforce = float(rng.choice([0, 0, 0.1])) if sim != 'ang' else 0.0
    wave = str(rng.choice(WAVES))
    precise = bool(rng.integers(0, 2))        # [r5] the split-precision ViT forward (the CLI's default) or f16 operands everywhere (drawn last: the other fields of a seed stay what they were)
    case = dict(precise=precise, H=H, W=W, S=S, sim=sim, opt=opt, align=align, kind=kind, fast=fast, sharp=sharp, expand=expand, enforce=enforce, wave=wave)
    case.update(force or {})
    if case['sim'] == 'ang':
        case['expand'] = case['enforce'] = 0.0
    if case['kind'] != 'dwt':
        case['wave'] = None
    return case


def run_case(model, case, seed, dev='cuda', tol=2e-3, steps=2):
    """-> list of (got, want) per step; raises AssertionError past `tol`"""
    cfg, wts = model.visual.cfg, model.visual.weights
    enc = lambda x: clip_vit_ref.encode_image(wts, x, cfg)
    H, W, S, kind, fast = case['H'], case['W'], case['S'], case['kind'], case['fast']
    sim, opt, align = case['sim'], case['opt'], case['align']
    sharp, expand, enforce = case['sharp'], case['expand'], case['enforce']
    exact_prev = transforms._EXACT_ZERO_ROT
    transforms._EXACT_ZERO_ROT = True          # 0-degree rotations through the bilinear pass, like the oracle's generic path
    try:
        seed_all(seed)
        tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
        trf = transforms.transforms_fast if fast else transforms.normalize()
        kw = dict(sim=sim, optimizer=opt, align=align, macro=0.4, sharp=sharp, expand=expand, enforce=enforce, transform=trf, rng='reference', precise=bool(case.get('precise', False)))
        okw = dict(sim=sim, optimizer=opt, align=align, sharp=sharp, expand=expand, enforce=enforce)
        if kind == 'dwt':
            from aphantasia_amd.image import dwt_image
            params, image_f, _ = dwt_image([1, 3, H, W], case['wave'], 0.3, 1.8, None)
            Ys = [p.detach().cpu().clone() for p in params]
            eng = Engine(image_f.flat.detach().clone(), H, W, model, S, [(tgt, -1.0)], param_kind='dwt', dwt=image_f.synth, **kw)
            H, W = eng.h, eng.w                # the synthesised frame may be a row / column larger (image.py:57)
            run = R.ReferenceRun(H, W, enc, [(tgt, 1.0)], params=Ys, param_kind='dwt', wave=case['wave'], dwt_sharp=0.3, **okw)
        else:
            p0 = R.fft_params_init([1, 3, H, W]).contiguous() if kind == 'fft' else torch.randn(1, 3, H, W) * 0.5
            if kind == 'pixel':
                kw.update(param_kind='pixel'); okw.update(param_kind='pixel')
            eng = Engine(p0.clone().to(dev).contiguous(), H, W, model, S, [(tgt, -1.0)], **kw)
            run = R.ReferenceRun(H, W, enc, [(tgt, 1.0)], params=p0, **okw)
        per = lambda augs: None if augs is None else (lambda c, cut: augment_ref.apply_fast(cut, augs[c], R.normalize))
        out = []
        for st in range(steps):
            seed_all(100 + st)
            tb, augs = draw_crop_params(S, 224, H, W, align, 0.4, trf)
            tb2 = augs2 = None
            if enforce != 0:
                tb2, augs2 = draw_crop_params(S, 224, H, W, align, 0.4, trf)
            want = run.step(tb, per(augs), tb2, per(augs2))
            got = float(eng.step(tb, augs, tables2=None if tb2 is None else (tb2, augs2)))
            if expand > 0:
                eng.set_prev_enc()
            out.append((got, want))
            assert abs(got - want) < tol, (st, got, want)
        assert int(eng.guard[0]) == 0, 'skipped step'
        return out
    finally:
        transforms._EXACT_ZERO_ROT = exact_prev


def run_seed(model, seed, cases, force=None, dev='cuda', verbose=True):
    """-> (list of bad (case, kind, message), worst |d loss|)"""
    rng = np.random.default_rng(seed)
    bad, worst, t0 = [], 0.0, time.time()
    for it in range(cases):
        case = draw_case(rng, force)
        try:
            res = run_case(model, case, it + 1000 * seed, dev)
            worst = max([worst] + [abs(g - w) for g, w in res])
            if verbose:
                print(it, case, 'ok  max |d loss| %.1e  %.0fs' % (max(abs(g - w) for g, w in res), time.time() - t0), flush=True)
        except AssertionError as e:
            bad.append((case, 'assert', str(e)[:120]))
            print(it, case, 'BAD', e, flush=True)
        except Exception as e:     # noqa: BLE001 -- a fuzz run reports every failure kind
            bad.append((case, type(e).__name__, str(e)[:200]))
            print(it, case, 'BAD', type(e).__name__, e, flush=True)
    return bad, worst
