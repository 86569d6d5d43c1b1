"""Launch-shape sweep of the separable crop adjoint (crop_adjoint_rows_kernel) at the headline size: rows per workgroup x columns per thread x
column segments x cuts per batch x row-block order, HIP-event timed through aph_sample_bwd with the same crop table.  Needs a
-DAPH_EXPERIMENTS build for the shapes the product library does not carry (`python -m aphantasia_amd._build --experiments`).
    python tools/exp/crop_adjoint_sweep.py [S] [H] [W] [patch]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi, ops, transforms
from aphantasia_amd.utils import draw_crop_params_bulk
S = int(sys.argv[1]) if len(sys.argv) > 1 else 190
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
P = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = 'cuda'
L = _ffi.lib()
rng = np.random.default_rng(0)
geom = ops.make_geom(H, W, S, 224, P)
table, _ = draw_crop_params_bulk(S, 224, H, W, 'uniform', 0.4, transforms.normalize(), rng)
tb = torch.from_numpy(table).to(dev)
ws = ops.sample_ws(geom, False, dev)
npatch = (224 // P) ** 2
g = torch.randn(S * npatch, 3 * P * P, device=dev)
grgb = torch.empty(3, H, W, device=dev)


def run():
    ops.sample_bwd(geom, g, tb, None, ws, grgb, _ffi.APH_OUT_PATCH_F16)


def timeit(n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


L.cdll.aph_crop_adjoint_set_shape(0, 0, 0, 0, -1)
run(); torch.cuda.synchronize()
ref = grgb.clone()
base = timeit()
print('%dx%d, %d cuts, patch %d: automatic shape %.1f us (tap tables + adjoint; experiments build: %s)' % (W, H, S, P, base, L.experiments), flush=True)
res = []
for nseg in (1, 2, 3, 4):
    for rb in ((4, 8, 9, 12, 16) if H <= 1080 else (12, 16, 20, 24)):
        for cpt in (1, 2, 3):
            xw = ((W + nseg - 1) // nseg + 3) & ~3
            nthr = ((xw + cpt - 1) // cpt + 63) // 64 * 64
            if nthr > 768 or (nthr < 256 and cpt > 1):
                continue
            for nbc in (12, 6):
                for order in (0, 1):
                    L.cdll.aph_crop_adjoint_set_shape(rb, cpt, nbc, nseg, order)
                    try:
                        run(); torch.cuda.synchronize()
                    except RuntimeError as e:
                        print('rb %2d cpt %d nseg %d nbc %2d order %d: refused (%s)' % (rb, cpt, nseg, nbc, order, str(e)[:80]), flush=True)
                        continue
                    ok = torch.equal(grgb, ref)
                    err = (grgb - ref).abs().max().item()
                    t = timeit()
                    res.append((t, rb, cpt, nseg, nbc, order, ok, err))
                    print('rb %2d cpt %d nseg %d nbc %2d order %d  (%3d threads, %4d workgroups): %7.1f us   %s' % (rb, cpt, nseg, nbc, order, max(nthr, 256), -(-H // rb) * 3 * nseg, t, 'bit-identical' if ok else 'max |diff| %.2e' % err), flush=True)
L.cdll.aph_crop_adjoint_set_shape(0, 0, 0, 0, -1)
print('best:')
for r in sorted(res)[:8]:
    print('  %7.1f us  rb %2d cpt %d nseg %d nbc %2d order %d  %s' % (r[0], r[1], r[2], r[3], r[4], r[5], 'bit-identical' if r[6] else 'max |diff| %.2e' % r[7]))
