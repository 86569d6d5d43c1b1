"""CPU: the product's HIP kernel sources executed under the host SIMT interpreter (tests/emu) through
the C ABI, checked against the oracle / reference-generated goldens (bodies in kernel_checks.py).
This is a logic check of the kernels (indexing, LDS staging, barriers, reductions); numerics on real
gfx950 hardware are covered by tests/test_gpu_kernels.py, which runs the same checks against
libaphantasia_hip.so."""
import os
import sys

import pytest
import numpy as np
import torch

from aphantasia_amd import _ffi
import kernel_checks as K

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))


@pytest.fixture(scope='module')
def emu():
    import build_emu
    return _ffi.Library(build_emu.build())


@pytest.mark.parametrize('name', ['synth_48x80.npz', 'synth_45x63.npz'])
def test_synth_golden(emu, golden, name):
    K.check_synth_golden(emu, 'cpu', golden(name))


def test_synth_oracle_shift_and_odd_radix(emu):
    K.check_synth_vs_oracle(emu, 'cpu', 24, 40, 1.0, with_shift=True)
    K.check_synth_vs_oracle(emu, 'cpu', 21, 26, 1.1)      # radix 3,7 / 2,13: generic butterflies


def test_dwt(emu):
    K.check_dwt(emu, 'cpu', 'db3', 45, 70)          # odd sizes: the dropped row/col path
    K.check_dwt(emu, 'cpu', 'coif2', 64, 96)
    K.check_dwt(emu, 'cpu', 'db3', 48, 72)          # rows of 4 outputs per lane (16-byte stores) in the finest level
    K.check_dwt(emu, 'cpu', 'db2', 40, 56)
    K.check_dwt(emu, 'cpu', 'db5', 50, 60)          # run-time filter length: the generic row-loop loads
    K.check_dwt(emu, 'cpu', 'haar', 40, 56)         # the L = 2 and L = 8 instantiations of the coarse-tail kernels
    K.check_dwt(emu, 'cpu', 'db4', 64, 80)
    K.check_dwt(emu, 'cpu', 'db3', 200, 300)        # two per-level launches (several tiles each), then the coarse tail


def test_fft_pair(emu):
    K.check_fft_pair(emu, 'cpu', 24, 40)
    K.check_fft_pair(emu, 'cpu', 21, 27)
    K.check_fft_pair(emu, 'cpu', 74, 51)      # 2 x 37 (a prime factor > 31: the direct-sum pass), 3 x 17
    K.check_synth_vs_oracle(emu, 'cpu', 41, 86, 1.1, with_shift=True)      # 41 prime, 2 x 43


def test_synth_spatial(emu):
    K.check_synth_spatial(emu, 'cpu')


@pytest.mark.parametrize('align', ['uniform', 'overscan', 'overmax'])
def test_sampler_golden(emu, golden, align):
    K.check_sampler_golden(emu, 'cpu', golden('slice_48x80.npz'), align)


@pytest.mark.parametrize('align,mode', [('uniform', _ffi.APH_OUT_NCHW_NORM), ('overscan', _ffi.APH_OUT_NCHW_RAW),
                                        ('uniform', _ffi.APH_OUT_PATCH_F16)])
def test_sampler_adjoint(emu, align, mode):
    K.check_sampler_adjoint(emu, 'cpu', align, mode)


def test_sampler_adjoint_column_segments(emu):
    """frames wider than 2304 columns: the separable crop adjoint runs one workgroup per (row block, channel, column SEGMENT); a segment culls
    the cuts that do not reach it (C4's 3840-wide frame is two segments)"""
    K.check_sampler_adjoint(emu, 'cpu', 'uniform', _ffi.APH_OUT_NCHW_RAW, H=30, W=2500, S=6, size=16, patch=8)


def test_sampler_augment(emu):
    K.check_sampler_augment(emu, 'cpu')
    K.check_sampler_augment(emu, 'cpu', H=80, W=96, S=6, size=64, patch=16)      # full 32x32 tiles: LDS-staged patch-major emit


def test_augment_kernels_vs_pillow(emu):
    pytest.importorskip('PIL.Image')
    K.check_kernels_vs_pil(emu, 'cpu')


def test_augment_invariants(emu):
    K.check_augment_invariants(emu, 'cpu')


def test_sim_loss(emu, golden):
    K.check_sim_loss(emu, 'cpu', golden('sim.npz'))
    K.check_sim_loss_per_cut(emu, 'cpu')


def test_linear_head(emu):
    K.check_linear_head(emu, 'cpu')


def test_adam(emu):
    K.check_adam(emu, 'cpu')
    K.check_adam(emu, 'cpu', n=5003)             # 16-byte quads + a scalar tail
    K.check_adam(emu, 'cpu', n=1001, offset=1)   # unaligned buffers: the scalar kernels


def test_gemm(emu):
    K.check_gemm(emu, 'cpu', [(100, 128, 64), (130, 256, 192)])
    K.check_gemm(emu, 'cpu', [(2100, 128, 192)])          # 256x128 tile config
    K.check_gemm(emu, 'cpu', [(520, 256, 128)])           # 64x64 tile config at a ragged M
    K.check_gemm(emu, 'cpu', [(150, 128, 512), (64, 128, 256)], tile_cfg=9)   # split-K x4, two-pass ordered reduction
    K.check_gemm(emu, 'cpu', [(70, 128, 192)], tile_cfg=8)
    K.check_gemm(emu, 'cpu', [(200, 256, 320)], tile_cfg=10)   # 128x128, 8 waves, 4-stage ring
    K.check_gemm(emu, 'cpu', [(70, 128, 64)], tile_cfg=2)     # single k-tile, single partial tile
    # wave-specialised persistent kernel (vit_gemm_ws.h): producer / consumer waves, permuted Bt rows, register epilogue;
    # single unit, ragged single tile, 9 / 10 tiles on 3 workgroups with odd and even k-tile counts
    K.check_gemm(emu, 'cpu', [(70, 128, 64), (300, 256, 192), (700, 768, 192), (1100, 512, 128)], tile_cfg=5, variants=(0,))
    # register-staged small-M kernels (vit_gemm_rs.h), step counts fixed at compile time per K.  Split-K (14 / 15: 4 / 3 k-steps in
    # flight): two to eight k-steps per wave (the two images and the register sets wrap), ragged M
    for cfg in (14, 15):
        K.check_gemm(emu, 'cpu', [(100, 128, 256), (130, 256, 2304), (70, 128, 768), (33, 384, 1024)], tile_cfg=cfg, variants=(0,))
    # A-resident (16 / 17: 8 / 4 weight k-steps in flight): as many k-steps as the prefetch depth (K = 256) and more, ragged M, two column groups
    for cfg in (16, 17):
        K.check_gemm(emu, 'cpu', [(100, 256, 256), (130, 512, 512), (70, 256, 768), (33, 512, 1024)], tile_cfg=cfg, variants=(0,))
    # panel-grouped tile order (groups of 2 / 4 row panels, ragged last group: 3 and 5 panels)
    for g in (2, 4):
        prev = emu.cdll.aph_gemm_set_ws_pgroup(g)
        try:
            K.check_gemm(emu, 'cpu', [(700, 768, 64), (1100, 512, 128), (70, 128, 64)], tile_cfg=5, variants=(0,))
        finally:
            emu.cdll.aph_gemm_set_ws_pgroup(prev)


def test_vit_through_the_wave_specialised_gemm(emu):
    """every ViT epilogue (f16 + bias, QuickGELU, its backward, fp32 residual with the accumulators started from it, patch embedding,
    f16 / f32 input gradient) on the wave-specialised kernel, forced at sizes the interpreter can run"""
    prev = emu.cdll.aph_gemm_set_ws_min_tiles(1)
    try:
        K.check_vit(emu, 'cpu', check_fuse=False)
        cfg = dict(input_resolution=64, patch_size=16, width=256, layers=2, heads=4, output_dim=128)      # T = 17
        K.check_vit(emu, 'cpu', cfg, S=20, check_fuse=False)        # M = 340: two row tiles, the second ragged
        # [r5] the split-precision QKV launch: long k-loops on the Q / K column tiles, half-length ones on the V column tiles, one walk
        # (6 column tiles x 2 row panels: 8 long + 4 short tiles over the interpreter's workgroups)
        K.check_vit(emu, 'cpu', cfg, S=20, check_fuse=False, hilo=True)
        for pg in (2,):                                             # another tile order of the same walk (row panels per group)
            prev_pg = emu.cdll.aph_gemm_set_ws_pgroup(pg)
            try:
                K.check_vit(emu, 'cpu', cfg, S=35, check_fuse=False, hilo=True)        # M = 595: three row panels, 12 long + 6 short tiles
            finally:
                emu.cdll.aph_gemm_set_ws_pgroup(prev_pg)
    finally:
        emu.cdll.aph_gemm_set_ws_min_tiles(prev)


def test_vit(emu):
    K.check_vit(emu, 'cpu')


def test_vit_split_precision_needs_enable_hilo(emu):
    """[r6] the K-repeated weight copies of the split-precision forward live in an arena of their own that only aph_vit_enable_hilo allocates:
    aph_vit_forward_hilo on a fresh handle is refused with a message that names the call; enabling is idempotent, survives a weight reload
    (the copies are refreshed), and the default forward never needs it"""
    import ctypes
    from aphantasia_amd import ops
    from aphantasia_amd.weights import synthetic_visual_weights
    cfg = K.TINY
    w = synthetic_visual_weights(cfg, 3)
    vit = ops.VitHandle(cfg, w, max_batch=2, lib=emu)
    base_bytes = vit.workspace_bytes()
    x = torch.randn(2, 3, cfg['input_resolution'], cfg['input_resolution'], generator=torch.Generator().manual_seed(1))
    p = cfg['patch_size']
    hilo = ops.patchify(x, p, lib=emu, hilo=True)
    out = torch.empty(2, cfg['output_dim'])
    rc = emu.cdll.aph_vit_forward_hilo(vit.handle, ops.ptr(hilo), 2, ops.ptr(out), None)
    assert rc < 0 and 'aph_vit_enable_hilo' in emu.last_error()
    enc_plain = vit.forward(ops.patchify(x, p, lib=emu), 2).clone()                 # the default path works on the bare handle
    vit.enable_hilo()
    vit.enable_hilo()                                                                # idempotent
    assert vit.workspace_bytes() > base_bytes
    enc = vit.forward(hilo, 2, hilo=True).clone()
    assert (enc - enc_plain).abs().max().item() < 3e-3 * enc_plain.abs().max().item()
    # a weight reload after enabling refreshes the copies: scaling in_proj by 0 on every layer changes the split forward exactly as the plain one
    for li in range(cfg['layers']):
        k = 'transformer.resblocks.%d.attn.in_proj_weight' % li
        a = np.ascontiguousarray((w[k] * 0.5).numpy())
        emu.call('aph_vit_set_weight', vit.handle, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size)
    enc2, enc2_plain = vit.forward(hilo, 2, hilo=True).clone(), vit.forward(ops.patchify(x, p, lib=emu), 2).clone()
    assert not torch.equal(enc2, enc) and (enc2 - enc2_plain).abs().max().item() < 3e-3 * enc2_plain.abs().max().item()


@pytest.mark.parametrize('fattn', [0, 2])
def test_vit_fused_forward_blocks(emu, fattn):
    """csrc/vit_block.h: LayerNorm inside the QKV / fc1 launches (DPP-row prologue into the resident A block), with (2) and without (0) the
    attention behind the (cut, head) QKV GEMM; T = 5 (one ragged tile), 17 (two row blocks of the flat kernel at S = 5) and 50 tokens.
    The same switch turns the fused BACKWARD on (a block's closing ln_1 input-gradient as the prologue of the next block's fc2 dgrad,
    fp32 stream handed over through a second buffer; the last block's class-rows-only residual included): check_vit covers both"""
    prev = emu.cdll.aph_vit_set_fused_max_rows(1 << 30)
    prev_a = emu.cdll.aph_vit_set_fused_attn(fattn)
    try:
        K.check_vit(emu, 'cpu', check_fuse=False)
        cfg = dict(input_resolution=64, patch_size=16, width=256, layers=2, heads=4, output_dim=128)      # T = 17
        K.check_vit(emu, 'cpu', cfg, S=5, check_fuse=False)
        cfg = dict(input_resolution=112, patch_size=16, width=256, layers=2, heads=4, output_dim=128)     # T = 50
        K.check_vit(emu, 'cpu', cfg, S=2, check_fuse=False)
    finally:
        emu.cdll.aph_vit_set_fused_max_rows(prev)
        emu.cdll.aph_vit_set_fused_attn(prev_a)


def test_vit_split_precision_forward(emu):
    """aph_vit_forward_hilo / aph_patchify_f16_hilo: [hi | lo] patch rows and first-LayerNorm outputs, GEMMs over twice the K; the saved
    activations serve the unchanged backward"""
    fe, be = K.check_vit(emu, 'cpu', hilo=True)
    cfg = dict(input_resolution=64, patch_size=16, width=256, layers=2, heads=4, output_dim=128)
    K.check_vit(emu, 'cpu', cfg, S=5, hilo=True)


def test_vit_50_tokens(emu):
    # T = 50 like ViT-B/32: four 16-row attention tiles with a ragged last one
    cfg = dict(input_resolution=112, patch_size=16, width=256, layers=1, heads=4, output_dim=128)
    K.check_vit(emu, 'cpu', cfg, S=2)


@pytest.mark.parametrize('res', [144, 224])
def test_vit_long_sequences(emu, res):
    # T = 82 (two 64-token blocks) and T = 197 like ViT-B/16 (four blocks, ragged last tile): blocked MFMA attention
    cfg = dict(input_resolution=res, patch_size=16, width=256, layers=1, heads=4, output_dim=128)
    K.check_vit(emu, 'cpu', cfg, S=1, check_fuse=False)      # (the fused LayerNorm pairs: test_vit and test_vit_50_tokens -- one block: first == last)


def test_rgb_priors(emu):
    K.check_rgb_priors(emu, 'cpu')


def test_rgb_sharp(emu):
    K.check_rgb_sharp(emu, 'cpu')


def test_rgb_to_u8(emu):
    K.check_rgb_to_u8(emu, 'cpu')


def test_frame_affine(emu):
    K.check_frame_affine(emu, 'cpu')


def test_adam_guard(emu):
    K.check_adam_guard(emu, 'cpu')


def test_depthwarp(emu, golden):
    K.check_depthwarp(emu, 'cpu', golden('depthwarp_40x56.npz'), sizes=((37, 51),))


def test_attention_alone(emu):
    K.check_attention(emu, 'cpu', S=5, T=50, heads=2)       # 10 (cut, head) items on 3 persistent workgroups: 4 / 3 / 3 items each
    K.check_attention(emu, 'cpu', S=1, T=60, heads=1)       # one item, 64-row tiles
    K.check_attention(emu, 'cpu', S=2, T=82, heads=1)       # blocked kernels (T > 64)
    K.check_attention(emu, 'cpu', S=4, T=70, heads=1)       # ... persistent: 4 items on 3 workgroups (one walks two items), one tile round
    K.check_attention(emu, 'cpu', S=2, T=135, heads=2)      # ... three 64-token blocks: two tile rounds per wave (fragment sets A / B), ragged last tile


def test_sampler_small_image_and_random_geometries(emu, monkeypatch):
    # regression: images under 16 pixels on a side (a 16 x 16 adjoint tile larger than the image) lost cuts in the adjoint's cull
    for align in ('uniform', 'overscan', 'overmax'):
        for H in (13, 15):
            K.check_sampler_adjoint(emu, 'cpu', align, _ffi.APH_OUT_NCHW_RAW, H=H, W=72, S=4, size=8, patch=8)
            K.check_sampler_adjoint(emu, 'cpu', align, _ffi.APH_OUT_PATCH_F16, H=72, W=H, S=4, size=8, patch=8)
    monkeypatch.setattr('aphantasia_amd.transforms._EXACT_ZERO_ROT', True)       # (the oracle resamples 0-degree cuts too)
    K.check_sampler_fuzz(emu, 'cpu', seed=11, n=14)
