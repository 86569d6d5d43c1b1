import os, sys, importlib.util
sys.path.insert(0, os.getcwd())
from aphantasia_amd import _ffi
spec = importlib.util.spec_from_file_location('loss_curve_tool', 'tools/loss_curve.py')
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
L = _ffi.lib()
for rs in (1, 0, 1, 0):
    L.cdll.aph_gemm_set_rs(rs)
    for precise in (False, True):
        worst, first, rms, got = tool.run_fixture('c2_s32_stress', precise=precise)
        print('rs=%d precise=%s: max |d loss| %.3e first past 1e-3 %s rms %.4f' % (rs, precise, worst, first, rms), flush=True)
    worst, first, rms, _ = tool.run_fixture('c2_s200', precise=True)
    print('rs=%d plain weights 200 cuts 50 steps precise: %.3e' % (rs, worst), flush=True)
