"""Round 3: fused first-block LayerNorm pairs (aph_vit_set_fuse_ln 1) against the separate kernels (0) on the GPU: how far apart are the
encodings and the patch gradient, where, and is either mode reproducible run to run?  (Under the host interpreter the two are equal bit for bit.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aphantasia_amd import _ffi, ops
from aphantasia_amd.weights import synthetic_visual_weights, visual_config
L = _ffi.lib()
for name, S in (('ViT-B/32', 4), ('ViT-B/16', 2)):
    cfg = visual_config(name)
    w = synthetic_visual_weights(cfg, 3)
    Rr, p = cfg['input_resolution'], cfg['patch_size']
    x = torch.randn(S, 3, Rr, Rr, generator=torch.Generator().manual_seed(1)).cuda()
    genc = (torch.randn(S, cfg['output_dim'], generator=torch.Generator().manual_seed(2)) * 0.01 * 1024).cuda()
    vit = ops.VitHandle(cfg, w, max_batch=S + 1)
    patches = ops.patchify(x.contiguous(), p)
    def run(mode):
        prev = L.call('aph_vit_set_fuse_ln', mode)
        try:
            e = vit.forward(patches, S).clone()
            g = vit.backward(genc, S, out_scale=1.0 / 1024).clone()
        finally:
            L.call('aph_vit_set_fuse_ln', prev)
        torch.cuda.synchronize()
        return e, g
    e1, g1 = run(1); e1b, g1b = run(1); e0, g0 = run(0); e0b, g0b = run(0)
    def rep(tag, a, b):
        d = (a.float() - b.float()).abs()
        nz = int((d > 0).sum())
        rows = torch.nonzero(d.reshape(a.shape[0], -1).amax(1) > 0).flatten().tolist()[:12]
        print('%s %-22s max|d| %.3e (max|a| %.3e)  differing %d / %d  finite %s  first rows %s' % (name, tag, d.max().item(), a.float().abs().max().item(), nz, d.numel(),
              bool(torch.isfinite(a.float()).all() and torch.isfinite(b.float()).all()), rows))
    rep('enc fused vs fused', e1, e1b); rep('grad fused vs fused', g1, g1b)
    rep('enc sep vs sep', e0, e0b); rep('grad sep vs sep', g0, g0b)
    rep('enc fused vs sep', e1, e0); rep('grad fused vs sep', g1, g0)
