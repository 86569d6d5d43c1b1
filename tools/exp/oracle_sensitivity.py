"""Round 3 (CPU only): how sensitive is the reference's own free-running loss curve to gradient perturbations of the size fp16
arithmetic produces?  The fp32 torch-CPU oracle is run again on the loss-curve fixture's configuration with a relative Gaussian
perturbation eps on every spectrum-gradient element per step, and compared with the unperturbed fixture
(tests/golden/loss_curve_<name>.npz).  If a 1e-6 perturbation already moves the curve by 1e-3, no reduced-precision path can track it.

    python tools/exp/oracle_sensitivity.py c2_s32_stress 30 1e-6 1e-4 1e-3          (multiplicative: g *= 1 + eps randn)
    python tools/exp/oracle_sensitivity.py c2_s32_stress 60 a1e-4 a1e-3 0            (a<eps>: ADDITIVE, g += eps max|g| randn -- the error
                                                                                       model of an fp16 backward; 0 = plain re-run, which differs
                                                                                       from the fixture only by the thread count's summation order)
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clip_vit_ref, reference_path as R
from oracle.make_loss_curves import CONFIGS, weights_of, seed_all

name, steps = sys.argv[1], int(sys.argv[2])
eps_list = [(v.startswith('a'), float(v.lstrip('a'))) for v in sys.argv[3:]]
c = CONFIGS[name]
fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'loss_curve_%s.npz' % name))['loss']
cfg, wts = weights_of(c['weights'])
for additive, eps in eps_list:
    seed_all(0)
    p0 = R.fft_params_init([1, 3, c['h'], c['w']])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    run = R.ReferenceRun(c['h'], c['w'], lambda x: clip_vit_ref.encode_image(wts, x, cfg), [(target, 1.0)], params=p0)
    seed_all(9)
    g = torch.Generator().manual_seed(123)
    worst, first = 0.0, None
    line = ''
    for i in range(steps):
        table = R.draw_crop_table(c['S'], 224, c['h'], c['w'], 'uniform', 0.4)
        loss = run.loss(table)
        run.opt.zero_grad(); loss.backward()
        with torch.no_grad():
            gr = run.params.grad
            if additive: gr.add_(eps * gr.abs().max() * torch.randn(gr.shape, generator=g))
            elif eps: gr.mul_(1 + eps * torch.randn(gr.shape, generator=g))
        run.opt.step(); run.i += 1
        d = abs(float(loss.detach()) - fx[i])
        worst = max(worst, d)
        if d > 1e-3 and first is None: first = i
        line += '%d:%.1e ' % (i, d)
    print('%s %s eps %.0e (%d threads): max |d loss| over %d steps %.2e, first step past 1e-3: %s\n   %s' % (name, 'additive' if additive else 'multiplicative', eps, torch.get_num_threads(), steps, worst, first, line), flush=True)
