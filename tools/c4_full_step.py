"""BASELINE configs[3] at FULL size, one step against the oracle: 3840x2160 DWT (db3, 11 levels) parameteriser, ViT-B/16,
--samples 400 -> 95 cuts (-tf none for the oracle's pinned path), sim 'mix', Adam.  Loss, coefficient-gradient cosine / max relative
error and the parameters after the update vs oracle.ReferenceRun (fp32 torch-CPU; about a minute of host CPU at 32 threads).

    python tools/c4_full_step.py > profiles/rNN_c4_full_step.txt
"""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_num_threads(min(32, os.cpu_count() or 1))
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from aphantasia_amd.image import dwt_image
from oracle import clip_vit_ref, reference_path as R

h, w, S = 2160, 3840, int(os.environ.get('S', 95))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/16', seed=1, max_batch=S)
torch.manual_seed(0); np.random.seed(0)
params, image_f, _ = dwt_image([1, 3, h, w], 'db3', 0.3, 1.8, None)
Ys = [p.detach().cpu().clone() for p in params]
tgt = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
eng = Engine(image_f.flat.detach().clone(), h, w, model, S, [(tgt, -1.0)], transform=transforms.normalize(), param_kind='dwt',
             dwt=image_f.synth, rng='reference', use_graph=False)
cfg, wts = model.visual.cfg, model.visual.weights
run = R.ReferenceRun(eng.h, eng.w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(tgt, 1.0)], params=Ys, param_kind='dwt', wave='db3', dwt_sharp=0.3)
torch.manual_seed(5); np.random.seed(5)
table = R.draw_crop_table(S, 224, eng.h, eng.w, 'uniform', 0.4)
got = float(eng.step(table)); torch.cuda.synchronize()
t0 = time.time()
want = run.step(table)
t_cpu = time.time() - t0
g, r = eng.grad.reshape(-1).double().cpu(), run.grad_flat().double()
cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
rel = (g - r).abs().max().item() / r.abs().max().item()
dp = (eng.params.reshape(-1).cpu() - run.params_flat()).abs()
print('# C4 full size: %dx%d DWT db3 (%d coefficients), ViT-B/16 (seeded synthetic weights), %d cuts, -tf none, one train(i) step' % (eng.w, eng.h, g.numel(), S))
print('loss_hip %.7f  loss_oracle %.7f  abs_diff %.2e' % (got, want, abs(got - want)))
print('coefficient gradient: cosine %.7f, max |diff| / max |g| %.2e' % (cos, rel))
print('parameters after Adam: mean |d| %.2e, fraction moved the other way (|d| > 0.02 = 2 lr) %.2e' % (dp.mean().item(), (dp > 0.02).float().mean().item()))
print('oracle step: %.1f s of host CPU (%d threads); skipped steps on the GPU: %d' % (t_cpu, torch.get_num_threads(), int(eng.guard[0])))
assert abs(got - want) < 1e-3 and cos > 0.999
