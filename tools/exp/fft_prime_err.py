"""Error of the any-size FFT synth (direct-sum pass for prime factors > 31) against the torch-CPU oracle as the prime grows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import kernel_checks as K
from aphantasia_amd import ops
from oracle import reference_path as R
dev = 'cuda'
for (h, w) in [(64, 74), (64, 202), (125, 1018), (125, 1307), (401, 1327), (1307, 128), (125, 1280), (127, 2003)]:
    for with_shift in (False, True):
        K.seed_all(1)
        params = R.fft_params_init([1, 3, h, w]).requires_grad_(True)
        scale = R.fft_scale(h, w, 1.5); cc_t = R.colcorr_t(1.8)
        shift = 0.02 * torch.rand(1, 1, h, w // 2 + 1, 1) if with_shift else None
        want = R.synth_fft(params, scale, h, w, cc_t, 1.1, shift)
        gw = torch.randn(1, 3, h, w)
        (want * gw).sum().backward()
        plan = ops.SynthPlan(3, h, w)
        sh = shift.reshape(h, w // 2 + 1).to(dev).contiguous() if with_shift else None
        raw, rgb = ops.synth_fft_fwd(plan, params.detach().to(dev).contiguous(), scale.to(dev), sh, 1.1, cc_t.flatten().tolist(), True)
        e1 = (rgb.cpu() - want.detach()[0]).abs().max().item()
        grad = ops.synth_fft_bwd(plan, gw[0].to(dev).contiguous(), rgb, raw, scale.to(dev), 1.1, cc_t.flatten().tolist(), True)
        ref = params.grad[0]
        e2 = (grad.cpu() - ref).abs().max().item() / ref.abs().max().item()
        print('%5d x %5d shift %d: rgb abs err %.2e (tol 4e-6)   grad rel err %.2e (tol 3e-5)' % (h, w, with_shift, e1, e2), flush=True)
