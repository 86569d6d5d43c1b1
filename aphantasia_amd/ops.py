"""Tensor-level wrappers over the C ABI (one Python function per aph_* entry point).

Every function runs the HIP library on the tensors' device memory on torch's current stream.
`lib=None` (the product path) uses aphantasia_amd/_ffi.lib() and insists on CUDA tensors; the
`lib` argument exists so the C-ABI can be driven through another loaded handle of the same ABI
(tests/emu runs the same kernel sources under a host interpreter -- test infrastructure only).
"""
import ctypes
import math
from ctypes import c_void_p, byref

import numpy as np
import torch

from . import _ffi
from ._ffi import SampleGeom, ptr, floats


def _L(lib, *tensors):
    if lib is not None:
        return lib
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('aphantasia_amd: the HIP path needs CUDA (ROCm) tensors; got a %s tensor. '
                               'There is no CPU implementation.' % t.device)
    return _ffi.lib()


def _stream(t):
    if t is not None and t.is_cuda:
        return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return c_void_p(0)


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError('%s must be a contiguous %s tensor' % (name, dtype))
    return t


# ------------------------------------------------------------------ parameteriser
class SynthPlan:
    def __init__(self, C, H, W, lib=None):
        self.lib = lib if lib is not None else _ffi.lib()
        self.C, self.H, self.W = C, H, W
        h = c_void_p()
        self.lib.call('aph_synth_plan_create', C, H, W, byref(h))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.cdll.aph_synth_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def synth_fft_fwd(plan, params, scale, shift=None, contrast=1.0, colcorr=None, decorrelate=True, lib=None):
    """params [1,3,H,Wc,2] or [3,H,Wc,2] f32 -> (raw [3,H,W], rgb [3,H,W])"""
    L = _L(lib, params, scale)
    _chk(params, torch.float32, 'params'); _chk(scale, torch.float32, 'scale')
    raw = torch.empty(plan.C, plan.H, plan.W, dtype=torch.float32, device=params.device)
    rgb = torch.empty_like(raw)
    L.call('aph_synth_fft_fwd', plan.handle, ptr(params), ptr(scale), ptr(shift), float(contrast), floats(colcorr),
           int(bool(decorrelate)), ptr(raw), ptr(rgb), _stream(params))
    return raw, rgb


def synth_fft_bwd(plan, d_rgb, rgb, raw, scale, contrast=1.0, colcorr=None, decorrelate=True, gscale=1.0, out=None, lib=None):
    L = _L(lib, d_rgb, rgb, raw, scale)
    _chk(d_rgb, torch.float32, 'd_rgb')
    if out is None:
        out = torch.empty(plan.C, plan.H, plan.W // 2 + 1, 2, dtype=torch.float32, device=d_rgb.device)
    L.call('aph_synth_fft_bwd', plan.handle, ptr(d_rgb), float(gscale), ptr(rgb), ptr(raw), ptr(scale), float(contrast),
           floats(colcorr), int(bool(decorrelate)), ptr(out), _stream(d_rgb))
    return out


def irfft2(plan, spectrum, out=None, lib=None):
    """[..,C,H,Wc,2] f32 -> [C,H,W] == torch.fft.irfftn(view_as_complex(x), s=(H,W), norm='ortho') (illustrip.py:401-403)"""
    L = _L(lib, spectrum)
    _chk(spectrum, torch.float32, 'spectrum')
    if spectrum.numel() != plan.C * plan.H * (plan.W // 2 + 1) * 2:
        raise ValueError('irfft2: %s does not match the plan (%d,%d,%d)' % (tuple(spectrum.shape), plan.C, plan.H, plan.W))
    if out is None:
        out = torch.empty(plan.C, plan.H, plan.W, dtype=torch.float32, device=spectrum.device)
    L.call('aph_irfft2', plan.handle, ptr(spectrum), ptr(out), _stream(spectrum))
    return out


def rfft2(plan, image, out=None, lib=None):
    """[C,H,W] f32 -> [C,H,Wc,2] == view_as_real(torch.fft.rfftn(x, s=(H,W), dim=[-2,-1], norm='ortho')) (illustrip.py:407-408)"""
    L = _L(lib, image)
    _chk(image, torch.float32, 'image')
    if image.numel() != plan.C * plan.H * plan.W:
        raise ValueError('rfft2: %s does not match the plan (%d,%d,%d)' % (tuple(image.shape), plan.C, plan.H, plan.W))
    if out is None:
        out = torch.empty(plan.C, plan.H, plan.W // 2 + 1, 2, dtype=torch.float32, device=image.device)
    L.call('aph_rfft2', plan.handle, ptr(image), ptr(out), _stream(image))
    return out


def synth_spatial_fwd(plan, raw, contrast=1.0, fixed_div=0.0, colcorr=None, decorrelate=True, lib=None):
    L = _L(lib, raw)
    _chk(raw, torch.float32, 'raw')
    rgb = torch.empty(plan.C, plan.H, plan.W, dtype=torch.float32, device=raw.device)
    L.call('aph_synth_spatial_fwd', plan.handle, ptr(raw), float(contrast), float(fixed_div), floats(colcorr),
           int(bool(decorrelate)), ptr(rgb), _stream(raw))
    return rgb


def synth_spatial_bwd(plan, d_rgb, rgb, raw, contrast=1.0, fixed_div=0.0, colcorr=None, decorrelate=True, gscale=1.0, lib=None):
    L = _L(lib, d_rgb, rgb, raw)
    _chk(d_rgb, torch.float32, 'd_rgb')
    out = torch.empty(plan.C, plan.H, plan.W, dtype=torch.float32, device=d_rgb.device)
    L.call('aph_synth_spatial_bwd', plan.handle, ptr(d_rgb), float(gscale), ptr(rgb), ptr(raw), float(contrast), float(fixed_div),
           floats(colcorr), int(bool(decorrelate)), ptr(out), _stream(d_rgb))
    return out


# ------------------------------------------------------------------ sampler
def make_geom(H, W, S, size, patch=32, align='uniform'):
    """aph_sample_geom for one slice_imgs call (padding per utils.py:232-236 / :178-186)."""
    Hp, Wp = H, W
    if 'over' in align:
        if align == 'overmax':
            Hp, Wp = 2 * H, 2 * W
        else:
            Hp, Wp = int(1.5 * H), int(1.5 * W)
    return SampleGeom(H, W, Hp, Wp, (Hp - H) // 2, (Wp - W) // 2, S, size, patch)


def sample_out_shape(geom, out_mode):
    if out_mode in (_ffi.APH_OUT_PATCH_F16, _ffi.APH_OUT_PATCH_F16_HILO):
        g = geom.size // geom.patch
        kx = 2 if out_mode == _ffi.APH_OUT_PATCH_F16_HILO else 1           # rows [hi | lo] of the split-precision forward
        return (geom.S * g * g, kx * 3 * geom.patch * geom.patch), torch.float16
    return (geom.S, 3, geom.size, geom.size), torch.float32


def sample_ws(geom, with_aug, device, lib=None):
    """caller-owned workspace of one aph_sample_fwd / aph_sample_bwd pair (tap tables + augmentation scratch)"""
    L = lib if lib is not None else _ffi.lib()
    n = int(L.cdll.aph_sample_ws_bytes(byref(geom), int(bool(with_aug))))
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device)


def sample_fwd(geom, rgb, table, aug=None, tmp=None, out=None, out_mode=_ffi.APH_OUT_NCHW_NORM, lib=None):
    """rgb [3,H,W] f32, table int32 [S,3] (device), aug f32 [S,16] (device) or None"""
    L = _L(lib, rgb, table, aug)
    _chk(rgb, torch.float32, 'rgb'); _chk(table, torch.int32, 'table')
    shape, dtype = sample_out_shape(geom, out_mode)
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=rgb.device)
    if tmp is None:
        tmp = sample_ws(geom, aug is not None, rgb.device, L)
    L.call('aph_sample_fwd', byref(geom), ptr(rgb), ptr(table), ptr(aug), ptr(tmp), ptr(out), int(out_mode), _stream(rgb))
    return out


def sample_bwd(geom, gout, table, aug=None, tmp=None, out=None, out_mode=_ffi.APH_OUT_NCHW_NORM, gscale=1.0, lib=None):
    L = _L(lib, gout, table, aug)
    _chk(gout, torch.float32, 'gout')
    if out is None:
        out = torch.empty(3, geom.H, geom.W, dtype=torch.float32, device=gout.device)
    if tmp is None:
        tmp = sample_ws(geom, aug is not None, gout.device, L)
    L.call('aph_sample_bwd', byref(geom), ptr(gout), float(gscale), ptr(table), ptr(aug), ptr(tmp), ptr(out), int(out_mode),
           _stream(gout))
    return out


def patchify(x, patch, lib=None, hilo=False):
    """NCHW f32 -> the patch-embed GEMM operand; hilo: rows [hi | lo] for VitHandle.forward(..., hilo=True)"""
    L = _L(lib, x)
    _chk(x, torch.float32, 'x')
    S, _, R, _ = x.shape
    g = R // patch
    out = torch.empty(S * g * g, (2 if hilo else 1) * 3 * patch * patch, dtype=torch.float16, device=x.device)
    L.call('aph_patchify_f16_hilo' if hilo else 'aph_patchify_f16', ptr(x), S, R, patch, ptr(out), _stream(x))
    return out


def unpatchify(g, S, R, patch, gscale=1.0, lib=None):
    L = _L(lib, g)
    _chk(g, torch.float32, 'g')
    out = torch.empty(S, 3, R, R, dtype=torch.float32, device=g.device)
    L.call('aph_unpatchify_f32', ptr(g), S, R, patch, float(gscale), ptr(out), _stream(g))
    return out


# ------------------------------------------------------------------ ViT
class VitHandle:
    """aph_vit: device weights + activation arena for `max_batch` cuts."""

    def __init__(self, cfg, weights, max_batch, lib=None):
        self.lib = lib if lib is not None else _ffi.lib()
        self.cfg = dict(cfg)
        self.max_batch = int(max_batch)
        h = c_void_p()
        self.lib.call('aph_vit_create', cfg['input_resolution'], cfg['patch_size'], cfg['width'], cfg['layers'], cfg['heads'],
                      cfg['output_dim'], self.max_batch, byref(h))
        self.handle = h
        for k, v in weights.items():
            a = np.ascontiguousarray(v.detach().cpu().float().numpy())
            self.lib.call('aph_vit_set_weight', self.handle, k.encode(), a.ctypes.data_as(c_void_p), a.size)
        g = cfg['input_resolution'] // cfg['patch_size']
        self.P, self.T, self.Kp = g * g, g * g + 1, 3 * cfg['patch_size'] ** 2
        self._hilo = False

    def enable_hilo(self):
        """the K-repeated weight copies of the split-precision forward (aph_vit_enable_hilo: 85 MB at ViT-B/32, not carried by default)"""
        if not self._hilo:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('VitHandle.enable_hilo() allocates: call it before the step is captured into a graph')
            self.lib.call('aph_vit_enable_hilo', self.handle)
            self._hilo = True

    def workspace_bytes(self):
        return int(self.lib.cdll.aph_vit_workspace_bytes(self.handle))

    def forward(self, patches, S, out=None, hilo=False):
        """hilo: the opt-in split-precision forward (aph_vit_forward_hilo); `patches` then holds [hi | lo] rows (APH_OUT_PATCH_F16_HILO)"""
        if out is None:
            out = torch.empty(S, self.cfg['output_dim'], dtype=torch.float32, device=patches.device)
        want = (2 if hilo else 1) * self.Kp
        if patches.shape[-1] != want:
            raise ValueError('VitHandle.forward(hilo=%s): patch rows of %d halfs, expected %d' % (hilo, patches.shape[-1], want))
        if hilo:
            self.enable_hilo()
        self.lib.call('aph_vit_forward_hilo' if hilo else 'aph_vit_forward', self.handle, ptr(patches), int(S), ptr(out), _stream(patches))
        return out

    def backward(self, genc, S, out=None, out_scale=1.0):
        if out is None:
            out = torch.empty(S * self.P, self.Kp, dtype=torch.float32, device=genc.device)
        fn = 'aph_vit_backward_h' if out.dtype == torch.float16 else 'aph_vit_backward'        # f16 `out`: patch gradient kept in half
        self.lib.call(fn, self.handle, ptr(genc), int(S), ptr(out), float(out_scale), _stream(genc))
        return out

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.cdll.aph_vit_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def gemm_f16(A, Bt, lib=None, tile_cfg=0):
    """C = A @ Bt^T (f16 in, f32 out); tile_cfg as in include/aphantasia_hip_test.h: 0 auto, 1 = 64x64, 2 = 256x128,
    4 = 256x256 phased, 8 / 9 = 64x64 split-K x2 / x4, 10 = 128x128, 22 / 24 = 128x128 split-K x2 / x4"""
    L = _L(lib, A, Bt)
    M, K = A.shape
    N = Bt.shape[0]
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    L.call('aph_gemm_f16_ld', ptr(A), K, ptr(Bt), K, M, N, K, ptr(C), int(tile_cfg), _stream(A))
    return C


# ------------------------------------------------------------------ loss / optimiser
def sim_loss(enc, targets, coef, sim_type='mix', denom=None, gscale=1.0, lib=None, per_sample=None, s_total=None, s_offset=0):
    """-> (loss [1] device tensor, genc [S,D]);  targets [T,D] broadcast embeddings, per_sample: optional
    [Tp, s_total, D] per-cut targets (coef lists broadcast ones first), coef: python floats (sign*weight)"""
    L = _L(lib, enc, targets)
    _chk(enc, torch.float32, 'enc')
    if targets is not None:
        _chk(targets, torch.float32, 'targets')
    S, D = enc.shape
    nb = 0 if targets is None else targets.shape[0]
    T = nb + (0 if per_sample is None else per_sample.shape[0])
    if per_sample is not None:
        s_total = per_sample.shape[1] if s_total is None else s_total
        flat = per_sample.reshape(-1, D).float()
        targets = flat.contiguous() if targets is None else torch.cat([targets, flat], 0).contiguous()
    code = _ffi.SIM_TYPES.get(sim_type if sim_type in _ffi.SIM_TYPES else _sim_key(sim_type))
    dcoef = torch.tensor(list(coef), dtype=torch.float32).to(enc.device)
    ws = torch.empty(S * (T + 2), dtype=torch.float32, device=enc.device)
    loss = torch.empty(1, dtype=torch.float32, device=enc.device)
    genc = torch.empty_like(enc)
    L.call('aph_sim_loss', ptr(enc), S, D, ptr(targets), ptr(dcoef), floats(list(coef)), T, nb, int(s_total or S), int(s_offset), code,
           float(S if denom is None else denom), float(gscale), ptr(ws), ptr(loss), ptr(genc), _stream(enc))
    return loss, genc


def _sim_key(t):
    """sim_func's substring dispatch (utils.py:277-295): 'mix' > 'spher' > 'ang' > 'dot' > cosine"""
    if t is None:
        return None
    if 'mix' in t:
        return 'mix'
    if 'spher' in t:
        raise ValueError("sim type 'spher' returns a per-sample vector upstream and cannot be used as a loss")
    if 'ang' in t:
        return 'ang'
    if 'dot' in t:
        return 'dot'
    return None


def adam_hyper(step, lr, beta1=0.0, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    return [lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, math.sqrt(1.0 - beta2 ** step), grad_scale]


def adam_step(p, g, m, v, vmax, hyper, decoupled=False, lib=None):
    """hyper: device f32 [8] (see adam_hyper)"""
    L = _L(lib, p, g, v, hyper)
    L.call('aph_adam_step', ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), ptr(hyper), int(bool(decoupled)), p.numel(), _stream(p))
