"""Randomised whole-step parity on the GPU (tests/engine_fuzz.py: random frame sizes, cut counts, similarity types, optimisers, --align
modes, FFT / pixel / DWT parameterisers, -tf none / fast, optional loss terms; two free-running steps of the fused engine against the
fp32 CPU oracle).  Fixed seeds of the same sweep run inside `pytest -m gpu` (test_engine_fuzz_seed).

    python tools/gpu_engine_fuzz.py [seed] [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import engine_fuzz as F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
bad, worst = F.run_seed(F.load_model(), seed, cases)
print('%d bad of %d in %.0f s (worst |d loss| %.1e)' % (len(bad), cases, time.time() - t0, worst))
