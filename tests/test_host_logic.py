"""CPU: host-side logic with no GPU -- the C-ABI library loads and exports every declared symbol, CLI
argument handling and the reference's sample-count derating, parameter layouts, weight loading."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def product_lib():
    from aphantasia_amd import _build, _ffi
    return _ffi.Library(_build.build(verbose=False))


def test_abi_exports_match_header(product_lib):
    from aphantasia_amd import _ffi
    hdr = open(os.path.join(ROOT, 'include', 'aphantasia_hip.h')).read()
    declared = set(re.findall(r'\b(aph_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for name in declared:
        assert hasattr(product_lib.cdll, name), name          # dlsym of every header symbol
    assert product_lib.cdll.aph_version() >= 100
    # error convention: negative code + message, no compute without a GPU
    assert product_lib.cdll.aph_synth_plan_create(3, 1, 1, None) < 0
    assert 'aph_synth_plan_create' in product_lib.last_error()


def test_product_path_refuses_cpu_tensors():
    from aphantasia_amd import ops
    with pytest.raises(RuntimeError, match='no CPU'):
        ops.gemm_f16(torch.zeros(4, 64).half(), torch.zeros(128, 64).half())


def test_cli_defaults_and_overrides():
    import clip_fft
    a = clip_fft.get_args(['-t', 'red square'])
    assert a.size == [720, 1280] and a.samples == 200 and a.steps == 200 and a.lrate == 0.05          # clip_fft.py:43,53-55,80
    assert (a.model, a.transform, a.optimizer, a.sim, a.align) == ('ViT-B/32', 'fast', 'adam_custom', 'mix', 'uniform')
    assert (a.contrast, a.colors, a.decay, a.macro) == (1.1, 1.8, 1.5, 0.4)
    a = clip_fft.get_args(['-t', 'x', '--size', '512'])
    assert a.size == [512, 512]                                                                         # :81
    a = clip_fft.get_args(['-t', 'x', '-dm', '3', '-m', 'ViT-B/16', '--sim', 'mix'])
    assert a.model == 'ViT-B/32' and a.sim == 'cossim'                                                  # :86-88
    a = clip_fft.get_args(['-t', 'x', '-r', 'snap.pt'])
    assert a.align == 'overscan'                                                                        # :82


@pytest.mark.parametrize('argv,want', [
    (['-t', 'x'], 190),                                        # C2: 200 * .95                       
    (['-t', 'x', '-tf', 'none'], 200),
    (['-t', 'x', '-dm', '2'], 43),                             # C3: int(200*.23)=46 -> int(46*.95)
    (['-t', 'x', '-m', 'ViT-B/16', '--samples', '400'], 95),   # C4: 400*.25=100 -> 95
    (['-t', 'x', '--samples', '1'], 0),                        # C1 with -tf fast collapses to 0 cuts upstream too
    (['-t', 'x', '-t2', 'y', '-t0', 'z'], 106),                # 190 -> int(142.5)=142 -> int(106.5)=106
    (['-t', 'x', '-e', '1', '-tf', 'none'], 100),
])
def test_sample_derating_matches_reference_arithmetic(argv, want):
    import clip_fft
    assert clip_fft.derate_samples(clip_fft.get_args(argv)) == want


def test_dwt_flat_layout_and_scale():
    from aphantasia_amd.dwt import coeff_shapes, dwt_scale_from_sizes, max_level
    from oracle import dwt_ref
    J, sizes = coeff_shapes(2160, 3840, 'db3')
    assert (J, sizes) == dwt_ref.coeff_shapes(2160, 3840, 'db3')
    Ys = dwt_ref.init_params([1, 3, 90, 120], 'coif2')
    _, sz = coeff_shapes(90, 120, 'coif2')
    assert dwt_scale_from_sizes(sz, 0.3) == dwt_ref.dwt_scale(Ys, 0.3)
    assert max_level(720, 1280) == 9


def test_checkpoint_loader_roundtrip(tmp_path):
    from aphantasia_amd.weights import load_openai_checkpoint, synthetic_visual_weights, visual_config
    cfg = visual_config('ViT-B/32')
    cfg['layers'] = 2
    w = synthetic_visual_weights(cfg, 0)
    sd = {'visual.' + k: v.half() for k, v in w.items()}
    sd['logit_scale'] = torch.tensor(1.0)
    path = os.path.join(tmp_path, 'ck.pt')
    torch.save(sd, path)
    vis, cfg2, full = load_openai_checkpoint(path)
    assert cfg2 == cfg and set(vis) == set(w)
    assert all(vis[k].dtype == torch.float32 for k in vis)


def test_alias_package_exposes_reference_names():
    import aphantasia.image as im
    import aphantasia.utils as ut
    import aphantasia.transforms as tr
    for n in ('to_valid_rgb', 'fft_image', 'dwt_image', 'pixel_image'):
        assert callable(getattr(im, n))
    for n in ('slice_imgs', 'sim_func', 'pad_up_to'):
        assert callable(getattr(ut, n))
    assert callable(tr.normalize) and tr.transforms_fast is not None
