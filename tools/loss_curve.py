"""Loss-vs-step of the HIP path against the fp32 CPU oracle ("loss curve within 1e-3"), free-running:
both sides take their own Adam steps from the same init with the same crop tables.  Writes a CSV.

    python tools/loss_curve.py [H W S STEPS] > profiles/rNN_loss_curve.csv
"""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphantasia_amd import clip as aclip, transforms
from aphantasia_amd.engine import Engine
from oracle import reference_path as R
from oracle import clip_vit_ref

h, w, S, steps = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (360, 640, 8, 10)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, _ = aclip.load('ViT-B/32', seed=1, max_batch=S)
torch.manual_seed(0); np.random.seed(0)
p0 = R.fft_params_init([1, 3, h, w])
target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
eng = Engine(p0.cuda().contiguous(), h, w, model, S, [(target, -1.0)], sim='mix', transform=transforms.normalize())
cfg, wts = model.visual.cfg, model.visual.weights
run = R.ReferenceRun(h, w, lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
torch.manual_seed(9); np.random.seed(9)
print('# %dx%d, ViT-B/32 (seeded synthetic weights), %d cuts, -tf none, sim mix, Adam(lr .05, b1 0); fp16-MFMA HIP path vs fp32 torch-CPU oracle' % (w, h, S))
print('step,loss_hip,loss_oracle,abs_diff')
worst = 0.0
for i in range(steps):
    table = R.draw_crop_table(S, 224, h, w, 'uniform', 0.4)
    got, want = float(eng.step(table)), run.step(table)
    worst = max(worst, abs(got - want))
    print('%d,%.7f,%.7f,%.2e' % (i, got, want, abs(got - want)))
with torch.no_grad():
    rms = (eng.synthesize(1.1).cpu() - run.image(1.1)[0]).pow(2).mean().sqrt().item()
print('# max |diff| %.2e ; final image pixel RMS (contrast 1.1) %.5f' % (worst, rms))
