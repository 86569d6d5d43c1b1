"""Round 3: where a tile's time goes inside the wave-specialised persistent GEMM (aph_gemm_ws_probe, shader-clock stamps of
consumer wave 0 of every workgroup): main loop vs epilogue per tile, with the ViT's real epilogues."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
M = int(os.environ.get('M', 9500))
KIND = {0: 'f16+bias', 1: 'gelu (2 x f16)', 2: 'f32 residual', 3: 'no store'}
CASES = [('qkv', 2304, 768, 0), ('qkv', 2304, 768, 3), ('fc1', 3072, 768, 1), ('fc1', 3072, 768, 3), ('outproj', 768, 768, 2), ('fc2', 768, 3072, 2), ('fc2', 768, 3072, 3)]
PAIR = 0
torch.manual_seed(0)
for (name, N, K, kind) in CASES:
    A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half()
    bias = torch.randn(N, device='cuda')
    o1 = torch.zeros(M, N, device='cuda', dtype=torch.float32 if kind in (2, 3) else torch.float16)
    o2 = torch.zeros(M, N, device='cuda', dtype=torch.float16)
    st = _stream(A)
    nwg = 512 if PAIR else 256
    tr = torch.zeros(nwg * 16 * 4, dtype=torch.int64, device='cuda')
    f = lambda t: L.call('aph_gemm_ws_probe', ptr(A), ptr(B), M, N, K, ptr(o1), ptr(o2), ptr(bias), kind | (0x10 if PAIR else 0), ptr(t) if t is not None else None, st)
    for _ in range(3): f(None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f(None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    f(tr); torch.cuda.synchronize()
    ti = tr.view(nwg, 16, 4).cpu()
    valid = ti[:, :, 1] != 0
    ntile = int(valid.sum(1).max())
    t = (ti - ti[:, :1, :1]).double()             # per workgroup, relative to its own first stamp (the XCDs' counters are not aligned)
    TICK = float(os.environ.get('TICK_US', 0.01)) # s_memtime: 100 MHz reference clock
    line = '%-8s N=%4d K=%4d %-15s %6.1f us %5.0f TF | tiles/wg %d |' % (name, N, K, KIND[kind] + (' PAIR' if PAIR else ''), us, 2.0 * M * N * K / us / 1e6, ntile)
    for j in range(ntile):
        ok = valid[:, j]
        k0, ml, ep = t[ok, j, 0] * TICK, t[ok, j, 1] * TICK, t[ok, j, 2] * TICK
        line += ' t%d: k0@%.1f main+%.1f epi+%.1f (max %.1f) end@%.1f (max %.1f) |' % (j, k0.mean().item(), (ml - k0).mean().item(), (ep - ml).mean().item(), (ep - ml).max().item(), ep.mean().item(), ep.max().item())
    # residency: entry / exit of every workgroup on the chip-wide 100 MHz clock -> how many are active at the same time
    ent, ext = ti[:, 15, 3], ti[:, 14, 3]
    t0_ = ent.min()
    ent, ext = (ent - t0_).double() * 0.01, (ext - t0_).double() * 0.01
    grid_t = torch.linspace(0, ext.max().item(), 200)
    conc = ((ent[None, :] <= grid_t[:, None]) & (ext[None, :] > grid_t[:, None])).sum(1)
    line += ' WG entry: median %.1f us max %.1f us after the first; WG duration median %.1f us; max concurrent WGs %d of %d |' % (
        ent.median().item(), ent.max().item(), (ext - ent).median().item(), int(conc.max()), nwg)
    print(line, flush=True)
