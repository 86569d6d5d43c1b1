"""Randomised parity sweep on the GPU (same checks as tests/kernel_checks.py, other seeds / larger shapes): sampler geometries,
attention shapes, GEMM shapes per tile configuration, FFT / DWT sizes.  python tools/gpu_fuzz.py [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import kernel_checks as K
from aphantasia_amd import transforms
transforms._EXACT_ZERO_ROT = True
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = []
t0 = time.time()


def attempt(name, f, *a, **k):
    try:
        f(*a, **k)
    except RuntimeError as e:
        if 'failed (-1)' in str(e) or 'failed (-3)' in str(e):      # a refused argument / unsupported shape: loud, not wrong
            return
        bad.append((name, a, k, 'RuntimeError', str(e)[:160]))
    except AssertionError as e:
        bad.append((name, a, k, 'assert', str(e)[:200]))


for s in range(3):
    attempt('sampler', K.check_sampler_fuzz, None, 'cuda', seed * 100 + s, 40)
attempt('sampler-large', K.check_sampler_fuzz, None, 'cuda', seed * 100 + 50, 16, max_hw=(500, 900))
for it in range(20):
    attempt('attention', K.check_attention, None, 'cuda', S=int(rng.integers(1, 40)), T=int(rng.integers(1, 257)), heads=int(rng.integers(1, 13)), seed=it)
for it in range(40):
    M = int(rng.integers(1, 12000)); N = 128 * int(rng.integers(1, 25)); Kk = 64 * int(rng.integers(1, 49))
    cfg = int(rng.choice([0, 1, 2, 5, 8, 9, 10, 11, 12, 22, 24]))
    attempt('gemm', K.check_gemm, None, 'cuda', [(M, N, Kk)], tile_cfg=cfg, variants=(0, 1) if cfg in (0, 1, 2, 10) else (0,))
for it in range(24):            # [r5] small batches: the register-staged split-K kernel (14 / 15) and what the heuristic picks for them (0)
    M = int(rng.integers(1, 200)); N = 64 * int(rng.integers(1, 49)); Kk = int(rng.choice([256, 768, 1024, 2304, 3072]))
    attempt('gemm-small', K.check_gemm, None, 'cuda', [(M, N, Kk)], tile_cfg=int(rng.choice([0, 14, 15])), variants=(0,))
for it in range(12):
    h = int(rng.integers(8, 900)); w = int(rng.integers(8, 1400))
    attempt('synth', K.check_synth_vs_oracle, None, 'cuda', h, w, 1.0 + 0.1 * (it % 2), with_shift=bool(it % 2))
    attempt('fft', K.check_fft_pair, None, 'cuda', h, w)
for it in range(6):
    attempt('dwt', K.check_dwt, None, 'cuda', str(rng.choice(['db2', 'db3', 'coif2', 'sym4', 'haar'])), int(rng.integers(32, 500)), int(rng.integers(32, 700)))
print('%d bad in %.0f s' % (len(bad), time.time() - t0))
for b in bad:
    print(b)
