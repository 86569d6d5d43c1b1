"""Where a workgroup's time goes inside the register-staged small-M GEMMs (aph_gemm_rs_probe: stamps of the chip-wide 100 MHz clock)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi
from aphantasia_amd.ops import ptr, _stream
L = _ffi.lib()
torch.manual_seed(0)
for M in [int(m) for m in os.environ.get('MS', '1200,2400').split(',')]:
    for (name, N, K, kind) in [('outproj', 768, 768, 0), ('fc2', 768, 3072, 0), ('dqkv', 768, 2304, 0), ('qkv', 2304, 768, 1), ('fc1', 3072, 768, 1), ('qkv packed', 2304, 768, 2), ('fc1 packed', 3072, 768, 2)]:
        A = torch.randn(M, K, device='cuda').half(); B = torch.randn(N, K, device='cuda').half()
        o = torch.empty(M, N, device='cuda', dtype=torch.float16)
        st = _stream(A)
        nwg = ((M + 63) // 64) * (N // (64 if kind == 0 else 256))
        tr = torch.zeros(nwg * 8, dtype=torch.int64, device='cuda')
        Bsrc = B
        if kind == 2:
            Bsrc = torch.empty_like(B)
            L.call('aph_gemm_pack_frag', ptr(B), N, K, ptr(Bsrc), st)
        f = lambda t: L.call('aph_gemm_rs_probe', ptr(A), ptr(Bsrc), M, N, K, ptr(o), kind, ptr(t) if t is not None else None, st)
        for _ in range(3): f(None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        f(tr); torch.cuda.synchronize()
        t = tr.view(nwg, 8).cpu().double() / 100.0          # us
        t0 = t[:, 0].min()
        ph = (t[:, 1:5] - t[:, 0:4])
        err = (o.float() - A.float() @ B.float().T).abs().max().item()
        print('%-10s err %.3f M %5d N %5d K %5d kind %d: %6.1f us/launch, %4d WGs | entry spread %.2f us, last exit at %.2f us | phases (mean us): %s | WG lifetime mean %.2f max %.2f'
              % (name, err, M, N, K, kind, us, nwg, (t[:, 0] - t0).max(), (t[:, 4] - t0).max(), ' '.join('%.2f' % v for v in ph.mean(0)), (t[:, 4] - t[:, 0]).mean(), (t[:, 4] - t[:, 0]).max()), flush=True)
