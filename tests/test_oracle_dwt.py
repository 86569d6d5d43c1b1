"""CPU: the DWT oracle restatement (oracle/dwt_ref.py) pinned against PyWavelets 1.1.1 run out-of-process
(/opt/conda/bin/python3.9, the only executable wavelet implementation in the build container)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

from oracle import dwt_ref

PY39 = '/opt/conda/bin/python3.9'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(PY39), reason='PyWavelets interpreter not present on this machine')
@pytest.mark.parametrize('wave,h,w', [('db3', 45, 70), ('coif2', 64, 96), ('haar', 33, 40), ('db2', 50, 37)])
def test_idwt_vs_pywt(wave, h, w):
    torch.manual_seed(0)
    Ys = [y.double() for y in dwt_ref.init_params([1, 3, h, w], wave)]
    img = dwt_ref.idwt(Ys[0], Ys[1:], wave)
    assert img.shape[-2] in (h, h + 1) and img.shape[-1] in (w, w + 1)
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, 'in.npz'), os.path.join(td, 'out.npy')
        d = dict(J=len(Ys) - 1, wave=wave, yl=Ys[0][0].numpy())
        for j in range(len(Ys) - 1):
            d['yh%d' % j] = Ys[j + 1][0].permute(1, 0, 2, 3).numpy()      # [3 bands, C, h, w]
        np.savez(inp, **d)
        subprocess.check_call([PY39, os.path.join(ROOT, 'oracle', 'pywt_dump.py'), 'waverec2', inp, out],
                              stderr=subprocess.DEVNULL)
        want = np.load(out)
    assert np.abs(img[0].numpy() - want).max() < 1e-10 * max(1.0, np.abs(want).max())


def test_shapes_match_survey_probe():
    # SURVEY.md section 2.3 (probed with pywt 1.1.1): 2160x3840 db3 -> 11 levels with these sizes
    J, sizes = dwt_ref.coeff_shapes(2160, 3840, 'db3')
    assert J == 11
    assert [s[0] for s in sizes] == [1082, 543, 274, 139, 72, 38, 21, 13, 9, 7, 6]
    assert [s[1] for s in sizes] == [1922, 963, 484, 244, 124, 64, 34, 19, 12, 8, 6]
    assert dwt_ref.max_level(720, 1280) == 9 and dwt_ref.max_level(224, 224) == 7
    assert (720 + 12 - 1) // 2 == 365


@pytest.mark.skipif(not os.path.exists(PY39), reason='PyWavelets interpreter not present on this machine')
@pytest.mark.parametrize('wave,h,w', [('db3', 45, 70), ('coif2', 64, 96), ('haar', 33, 40)])
def test_forward_dwt_host_vs_pywt_and_round_trip(wave, h, w):
    """aphantasia_amd.dwt.dwt_forward_host (img2dwt, image.py:82-94) == pywt.wavedec2(mode='symmetric') run out of process;
    and the oracle's inverse transform reconstructs the image from it (perfect reconstruction, band order included)"""
    from aphantasia_amd.dwt import dwt_forward_host, max_level
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 1, h, w, generator=g, dtype=torch.float64)
    J = max_level(h, w)
    yl, yh = dwt_forward_host(x, wave)
    assert len(yh) == J
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, 'in.npz'), os.path.join(td, 'out.npz')
        np.savez(inp, x=x[0, 0].numpy(), wave=wave, J=J)
        subprocess.check_call([PY39, os.path.join(ROOT, 'oracle', 'pywt_dump.py'), 'wavedec2', inp, out], stderr=subprocess.DEVNULL)
        want = np.load(out)
    assert np.abs(yl[0, 0].numpy() - want['yl']).max() < 1e-5
    for j in range(J):
        assert np.abs(yh[j][0, 0].numpy() - want['yh%d' % j]).max() < 1e-5, j
    rec = dwt_ref.idwt(yl.double(), [y.double() for y in yh], wave)
    assert (rec[..., :h, :w] - x).abs().max().item() < 1e-5
