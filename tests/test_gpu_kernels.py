"""GPU (MI355X): every kernel of libaphantasia_hip.so through the C ABI against the oracle and the
reference-generated goldens, at the golden sizes and at BASELINE.json's full sizes."""
import os

import pytest
import torch

from aphantasia_amd import _ffi, ops
import kernel_checks as K

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('name', ['synth_48x80.npz', 'synth_45x63.npz'])
def test_synth_golden(golden, name):
    K.check_synth_golden(None, DEV, golden(name))


def test_synth_oracle_small():
    K.check_synth_vs_oracle(None, DEV, 24, 40, 1.0, with_shift=True)
    K.check_synth_vs_oracle(None, DEV, 21, 26, 1.1)


def test_synth_oracle_full_size():
    K.check_synth_vs_oracle(None, DEV, 720, 1280, 1.0)


def test_fft_pair():
    K.check_fft_pair(None, DEV, 24, 40)
    K.check_fft_pair(None, DEV, 45, 63)
    K.check_fft_pair(None, DEV, 720, 1280)
    K.check_fft_pair(None, DEV, 768, 1366)    # 1366 = 2 x 683: a large prime factor (direct-sum pass), as torch.fft accepts any size
    K.check_synth_vs_oracle(None, DEV, 37 * 6, 101 * 4, 1.0)


def test_depthwarp_vs_reference_golden_and_oracle(golden):
    """depth.py:41-84 on the GPU: the reference's own outputs (golden) + the oracle at other sizes, incl. 720p"""
    K.check_depthwarp(None, DEV, golden('depthwarp_40x56.npz'), sizes=((37, 51), (64, 48), (720, 1280)))


def test_dwt():
    K.check_dwt(None, DEV, 'db3', 45, 70)
    K.check_dwt(None, DEV, 'coif2', 64, 96)
    K.check_dwt(None, DEV, 'haar', 96, 128)
    K.check_dwt(None, DEV, 'db3', 540, 960)          # quarter of BASELINE configs[3] (the CPU oracle stays quick)


def test_synth_spatial():
    K.check_synth_spatial(None, DEV)
    K.check_synth_spatial(None, DEV, 720, 1280)


@pytest.mark.parametrize('align', ['uniform', 'overscan', 'overmax'])
def test_sampler_golden(golden, align):
    K.check_sampler_golden(None, DEV, golden('slice_48x80.npz'), align)


@pytest.mark.parametrize('align,mode', [('uniform', _ffi.APH_OUT_NCHW_NORM), ('overscan', _ffi.APH_OUT_NCHW_RAW),
                                        ('uniform', _ffi.APH_OUT_PATCH_F16)])
def test_sampler_adjoint(align, mode):
    K.check_sampler_adjoint(None, DEV, align, mode)


def test_sampler_adjoint_full_size():
    K.check_sampler_adjoint(None, DEV, 'uniform', _ffi.APH_OUT_PATCH_F16, H=720, W=1280, S=12, size=224, patch=32)
    K.check_sampler_adjoint(None, DEV, 'overscan', _ffi.APH_OUT_NCHW_NORM, H=360, W=640, S=6, size=224, patch=32)
    # wider than one column segment of the separable adjoint (2304): two / three segments, patch-major and planar gradients
    K.check_sampler_adjoint(None, DEV, 'uniform', _ffi.APH_OUT_PATCH_F16, H=300, W=3840, S=8, size=224, patch=32)
    K.check_sampler_adjoint(None, DEV, 'central', _ffi.APH_OUT_NCHW_RAW, H=260, W=5000, S=6, size=224, patch=32)


def test_sampler_augment():
    K.check_sampler_augment(None, DEV)
    K.check_sampler_augment(None, DEV, H=360, W=640, S=12, size=224, patch=32)


def test_augment_invariants():
    K.check_augment_invariants(None, DEV)
    K.check_augment_invariants(None, DEV, size=224, patch=32)


def test_sim_loss(golden):
    K.check_sim_loss(None, DEV, golden('sim.npz'))
    K.check_sim_loss_per_cut(None, DEV)


def test_linear_head():
    K.check_linear_head(None, DEV)


def test_adam():
    K.check_adam(None, DEV)
    K.check_adam(None, DEV, n=5003)
    K.check_adam(None, DEV, n=1001, offset=1)
    K.check_adam(None, DEV, n=2769120)


def test_attention_alone_vs_torch():
    """attention forward / backward kernels through the test hook: ViT-B/32 (T = 50, full batch of 190 cuts x 12 heads: the persistent
    backward's item loop), T = 64 tiles, ViT-B/16 (T = 197) and the sizes in between"""
    K.check_attention(None, DEV, S=190, T=50, heads=12)
    K.check_attention(None, DEV, S=7, T=50, heads=12)
    for T in (17, 56, 57, 64, 82, 145, 197):
        K.check_attention(None, DEV, S=3, T=T, heads=12, seed=T)


def test_crop_adjoint_rows_kernel_vs_gather_kernel_full_size():
    """[r3] the separable row-block crop adjoint (frames without wrap padding) against the round-2 per-pixel gather kernel at the headline
    geometry (1280x720, 190 cuts, patch-major gradient; and the planar layout the -tf fast chain hands it), a small odd frame, and 4K width
    (W > 2304: the launcher keeps the gather kernel there); same bits on every launch -- the tripwire for the kernel's double-buffered tables"""
    import numpy as np
    import torch
    from aphantasia_amd import ops
    from aphantasia_amd.utils import draw_crop_params_bulk
    L = _ffi.lib()
    for (H, W, S, mode) in ((720, 1280, 190, _ffi.APH_OUT_PATCH_F16), (720, 1280, 48, _ffi.APH_OUT_NCHW_RAW), (2160, 3840, 24, _ffi.APH_OUT_PATCH_F16),
                            (300, 500, 16, _ffi.APH_OUT_NCHW_NORM)):
        rng = np.random.default_rng(H)
        geom = ops.make_geom(H, W, S, 224, 32)
        table, _ = draw_crop_params_bulk(S, 224, H, W, 'uniform', 0.4, None, rng)
        tb = torch.from_numpy(table).to(DEV)
        g = torch.randn(S * 49, 3072, device=DEV) if mode == _ffi.APH_OUT_PATCH_F16 else torch.randn(S, 3, 224, 224, device=DEV)
        new = ops.sample_bwd(geom, g, tb, out_mode=mode, gscale=0.5).clone()
        again = ops.sample_bwd(geom, g, tb, out_mode=mode, gscale=0.5)
        assert torch.equal(new, again)
        prev = L.cdll.aph_crop_adjoint_set_gather(1)
        try:
            old = ops.sample_bwd(geom, g, tb, out_mode=mode, gscale=0.5).clone()
        finally:
            L.cdll.aph_crop_adjoint_set_gather(prev)
        err = (new - old).abs().max().item() / old.abs().max().item()
        assert err < 2e-6, (H, W, S, mode, err)


def test_attention_backward_is_deterministic():
    import torch
    a1, d1 = K.check_attention(None, DEV, S=190, T=50, heads=12)
    a2, d2 = K.check_attention(None, DEV, S=190, T=50, heads=12)
    assert torch.equal(a1, a2) and torch.equal(d1, d2)


def test_sampler_random_geometries(monkeypatch):
    """ragged images / every align mode / every layout / with and without -tf fast, forward + adjoint vs the oracle"""
    monkeypatch.setattr('aphantasia_amd.transforms._EXACT_ZERO_ROT', True)       # (the oracle resamples 0-degree cuts too)
    K.check_sampler_fuzz(None, DEV, seed=5, n=60)
    K.check_sampler_fuzz(None, DEV, seed=6, n=12, max_hw=(400, 700))
    for H in (13, 15):
        K.check_sampler_adjoint(None, DEV, 'overscan', 0, H=H, W=72, S=4, size=8, patch=8)


def test_augment_kernels_vs_pillow():
    """perspective / rotation stages of the sampler and aph_frame_affine against Pillow's float Image.transform (interior pixels)"""
    pytest.importorskip('PIL.Image')
    K.check_kernels_vs_pil(None, DEV)


def test_augment_vs_torchvision_fixture():
    """the sampler's perspective / erase / rotate stages and aph_frame_affine against torchvision's outputs (skips while the fixture is absent)"""
    K.check_kernels_vs_tv_fixture(None, DEV, K.tv_fixture_or_skip())


def test_gemm_mfma_layout():
    K.check_gemm(None, DEV, [(100, 128, 64), (130, 256, 192), (9500, 768, 768), (1000, 3072, 768), (777, 768, 3072)])


def test_gemm_every_tile_config():
    for cfg in (1, 2, 10):
        K.check_gemm(None, DEV, [(9500, 768, 768), (333, 256, 64), (1200, 3072, 768)], tile_cfg=cfg)
    K.check_gemm(None, DEV, [(9500, 768, 768), (333, 256, 64)], tile_cfg=11, variants=(0,))       # two workgroups per CU (2-stage ring)


def test_gemm_wave_specialised_vs_matmul_and_reproducible():
    """tile_cfg 5 (vit_gemm_ws.h): every ViT-B shape at full batch and ragged / single-tile cases against fp32 matmul, and the SAME BITS
    on every launch -- the interpreter cannot see a vmcnt under-wait or a stage refilled too early; a race shows up here as a changing tile"""
    import torch
    from aphantasia_amd import ops
    K.check_gemm(None, DEV, [(9500, 2304, 768), (9500, 768, 768), (9500, 3072, 768), (9500, 768, 3072), (9500, 768, 2304), (70, 128, 64),
                             (333, 256, 64), (18715, 3072, 768)], tile_cfg=5, variants=(0,))
    g = torch.Generator().manual_seed(7)
    for (M, N, Kd) in ((9500, 2304, 768), (9500, 768, 3072), (4750, 3072, 768)):
        A = torch.randn(M, Kd, generator=g).half().to(DEV); Bt = torch.randn(N, Kd, generator=g).half().to(DEV)
        ref = ops.gemm_f16(A, Bt, tile_cfg=5).clone()
        busy = torch.randn(4096, 4096, device=DEV)
        for i in range(12):
            if i % 3 == 0:
                busy = busy * 1.0001          # uneven load next to the launches
            assert torch.equal(ops.gemm_f16(A, Bt, tile_cfg=5), ref), (M, N, Kd, i)
    # the tile ORDER does not change a tile's arithmetic: n-fastest (1) == groups of 2 / 4 / 5 row panels, bit for bit
    from aphantasia_amd import _ffi
    L = _ffi.lib()
    A = torch.randn(9500, 768, generator=g).half().to(DEV); Bt = torch.randn(3072, 768, generator=g).half().to(DEV)
    outs = []
    for pg in (1, 2, 4, 5):
        prev = L.cdll.aph_gemm_set_ws_pgroup(pg)
        try:
            outs.append(ops.gemm_f16(A, Bt, tile_cfg=5).clone())
        finally:
            L.cdll.aph_gemm_set_ws_pgroup(prev)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert (outs[0] - A.float() @ Bt.float().T).abs().max().item() < 2e-3 * (768 / 64) ** 0.5


def test_vit_forced_through_the_wave_specialised_gemm():
    """all of the ViT's epilogues on the wave-specialised kernel at small sizes (the default heuristic only takes shapes with >= 160 tiles)"""
    from aphantasia_amd import _ffi
    L = _ffi.lib()
    prev = L.cdll.aph_gemm_set_ws_min_tiles(1)
    try:
        K.check_vit(None, DEV)
        cfg = dict(input_resolution=64, patch_size=16, width=256, layers=2, heads=4, output_dim=128)
        K.check_vit(None, DEV, cfg, S=40)
    finally:
        L.cdll.aph_gemm_set_ws_min_tiles(prev)


def test_gemm_register_staged_small_m_vs_matmul_and_reproducible():
    """tile_cfg 14 / 15 (split-K; and 16 / 17, A-resident, in -DAPH_EXPERIMENTS builds), vit_gemm_rs.h: the small-M kernels -- every wave stages its own operand stream
    through registers and a private LDS image with no barrier in the main loop.  The ViT-B shapes at the shard sizes of 8 / 4 / 2 ranks,
    ragged and single-k-tile cases against fp32 matmul, and the SAME BITS on every launch next to uneven load"""
    import torch
    from aphantasia_amd import ops
    for cfg in (14, 15):
        K.check_gemm(None, DEV, [(1200, 768, 768), (1200, 768, 3072), (1150, 768, 2304), (2400, 768, 3072), (50, 768, 768), (50, 768, 3072),
                                 (70, 128, 256), (333, 256, 1024), (4750, 768, 2304)], tile_cfg=cfg, variants=(0,))
    for cfg in ((16, 17) if _ffi.lib().experiments else ()):
        K.check_gemm(None, DEV, [(1200, 2304, 768), (1200, 3072, 768), (24, 3072, 768), (50, 2304, 768), (70, 256, 256), (333, 512, 512),
                                 (4750, 2304, 768), (2150, 3072, 768), (100, 1024, 1024)], tile_cfg=cfg, variants=(0,))
    g = torch.Generator().manual_seed(11)
    for (M, N, Kd, cfgs) in ((1200, 768, 3072, (14, 15)), (2150, 768, 2304, (14, 15)), (50, 768, 3072, (14, 15))) + \
            (((1200, 2304, 768, (16, 17)), (50, 3072, 768, (16, 17))) if _ffi.lib().experiments else ()):
        A = torch.randn(M, Kd, generator=g).half().to(DEV); Bt = torch.randn(N, Kd, generator=g).half().to(DEV)
        for cfg in cfgs:
            ref = ops.gemm_f16(A, Bt, tile_cfg=cfg).clone()
            busy = torch.randn(4096, 4096, device=DEV)
            for i in range(12):
                if i % 3 == 0:
                    busy = busy * 1.0001          # uneven load next to the launches
                assert torch.equal(ops.gemm_f16(A, Bt, tile_cfg=cfg), ref), (M, N, Kd, cfg, i)


@pytest.mark.parametrize('fattn', [0, 2])
def test_vit_fused_forward_blocks(fattn):
    """csrc/vit_block.h on hardware: tiny and ViT-B/32-shaped models through the fused forward (LayerNorm inside the QKV / fc1 launches,
    with and without the attention behind the QKV GEMM) against the fp32 oracle, and bit-identical on repetition"""
    import torch
    from aphantasia_amd import _ffi, ops
    from aphantasia_amd.weights import synthetic_visual_weights
    L = _ffi.lib()
    if not L.experiments:
        pytest.skip('the fused block kernels were measured slower than the per-operator path at every shard size (profiles/r05_fused_v4_steps.txt) and are '
                    'compiled into -DAPH_EXPERIMENTS builds only; they passed on hardware in profiles/r05_gpu_tests_fused_blocks.log and run under '
                    'the CPU interpreter in tests/test_emu_kernels.py')
    prev = L.cdll.aph_vit_set_fused_max_rows(1 << 30)
    prev_a = L.cdll.aph_vit_set_fused_attn(fattn)
    try:
        K.check_vit(None, DEV, check_fuse=False)
        cfg = dict(input_resolution=112, patch_size=16, width=256, layers=2, heads=4, output_dim=128)     # T = 50
        K.check_vit(None, DEV, cfg, S=7, check_fuse=False)
        from aphantasia_amd.weights import visual_config
        cfg = visual_config('ViT-B/32')
        K.check_vit(None, DEV, cfg, S=5, check_fuse=False, fwd_tol=5e-3, bwd_tol=3e-2)
        w = synthetic_visual_weights(cfg, 3)
        vit = ops.VitHandle(cfg, w, max_batch=24)
        x = torch.randn(24, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(DEV)
        patches = ops.patchify(x, 32)
        ref = vit.forward(patches, 24).clone()
        for _ in range(5):
            assert torch.equal(vit.forward(patches, 24), ref)
    finally:
        L.cdll.aph_vit_set_fused_max_rows(prev)
        L.cdll.aph_vit_set_fused_attn(prev_a)


def test_gemm_splitk_matches_and_is_deterministic():
    """split-K (tile_cfg 8 / 9): ordered last-block reduction -> same bits on every run, fp32-rounding close to the unsplit kernel"""
    import torch
    from aphantasia_amd import ops
    K.check_gemm(None, DEV, [(1200, 768, 3072), (190, 768, 3072), (1200, 768, 2304), (77, 128, 512)], tile_cfg=9)
    K.check_gemm(None, DEV, [(2400, 768, 3072), (333, 256, 192)], tile_cfg=8)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(1200, 3072, generator=g).half().to(DEV); Bt = torch.randn(768, 3072, generator=g).half().to(DEV)
    ref = ops.gemm_f16(A, Bt, tile_cfg=9).clone()
    for _ in range(20):
        assert torch.equal(ops.gemm_f16(A, Bt, tile_cfg=9), ref)
    one = ops.gemm_f16(A, Bt, tile_cfg=1)
    assert (one - ref).abs().max().item() < 2e-3 * one.abs().max().item()


def test_vit_tiny():
    K.check_vit(None, DEV)


@pytest.mark.parametrize('res', [80, 144, 192, 224, 256])
def test_vit_sequence_lengths(res):
    # T = 26, 82, 145, 197, 257->rejected?  (one-tile MFMA attention, then the blocked kernels with 2, 3, 4 blocks)
    cfg = dict(input_resolution=res, patch_size=16, width=256, layers=2, heads=4, output_dim=128)
    if (res // 16) ** 2 + 1 > 256:
        with pytest.raises(RuntimeError):
            K.check_vit(None, DEV, cfg, S=2)
        return
    K.check_vit(None, DEV, cfg, S=2)


@pytest.mark.parametrize('name', ['ViT-B/32', 'ViT-B/16'])
def test_vit_base(name):
    from aphantasia_amd.weights import visual_config
    ferr, berr = K.check_vit(None, DEV, visual_config(name), S=3, fwd_tol=5e-3, bwd_tol=3e-2)
    print('%s fwd rel err %.2e  bwd rel err %.2e' % (name, ferr, berr))


def test_rgb_priors():
    K.check_rgb_priors(None, DEV)


def test_rgb_sharp():
    K.check_rgb_sharp(None, DEV)


def test_frame_affine():
    K.check_frame_affine(None, DEV)


def test_adam_guard():
    K.check_adam_guard(None, DEV)
