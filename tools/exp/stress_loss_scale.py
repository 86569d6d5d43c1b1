"""Does the stress-weight loss curve (tests/golden/loss_curve_c2_s32_stress.npz) depend on the static fp16 loss scale?
If the curve error came from gradients falling into fp16's subnormal range, a larger scale would shrink it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.loss_curve import run_fixture

for name in ('c2_s32_stress', 'c2_s32'):
    for ls in (64.0, 1024.0, 4096.0, 65536.0, 2.0 ** 20, 2.0 ** 24):
        try:
            mx, first, rms, got = run_fixture(name, steps=60, loss_scale=ls)
            print('%s loss_scale %g: max |d loss| over 60 steps %.2e, first step past 1e-3: %s' % (name, ls, mx, first), flush=True)
        except Exception as e:
            print('%s loss_scale %g: %s' % (name, ls, e), flush=True)
