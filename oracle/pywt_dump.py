"""TEST/BUILD INFRASTRUCTURE.  Runs under /opt/conda/bin/python3.9 (PyWavelets 1.1.1 -- the version the
reference's notebook pins; the main interpreter has no pywt).  Two jobs:
  python3.9 oracle/pywt_dump.py filters OUT.py      -> reconstruction filter taps for the wavelets the CLI accepts
  python3.9 oracle/pywt_dump.py waverec2 IN.npz OUT.npy -> pywt.waverec2 of dumped coefficients (DWT oracle pin)
  python3.9 oracle/pywt_dump.py wavedec2 IN.npz OUT.npz -> pywt.wavedec2 of a dumped image (forward-transform pin)
"""
import sys
import numpy as np
import pywt


def filters(out):
    names = ['haar', 'dmey'] + ['db%d' % i for i in range(1, 21)] + ['coif%d' % i for i in range(1, 11)] + ['sym%d' % i for i in range(2, 11)]
    with open(out, 'w') as f:
        f.write('"""Reconstruction low-pass filters (rec_lo) of the wavelets accepted by `--wave`, dumped from\n'
                'PyWavelets %s by oracle/pywt_dump.py.  rec_hi[k] = (-1)**k * rec_lo[L-1-k] (orthogonal QMF);\n'
                'dec_lo = rec_lo[::-1].  Data table, generated -- do not edit."""\n\nREC_LO = {\n' % pywt.__version__)
        for n in names:
            w = pywt.Wavelet(n)
            rl, rh = np.array(w.rec_lo), np.array(w.rec_hi)
            qmf = np.array([(-1) ** k * rl[len(rl) - 1 - k] for k in range(len(rl))])
            assert np.allclose(qmf, rh, atol=1e-12), n
            f.write('    %r: [%s],\n' % (n, ', '.join(repr(float(v)) for v in rl)))
        f.write('}\n')


def waverec2(inp, out):
    d = np.load(inp)
    J = int(d['J'])
    coeffs = [d['yl']]
    for j in range(J - 1, -1, -1):          # coarsest first for pywt
        yh = d['yh%d' % j]
        coeffs.append((yh[0], yh[1], yh[2]))
    np.save(out, pywt.waverec2(coeffs, str(d['wave']), 'symmetric'))


def wavedec2(inp, out):
    """forward transform pin: IN.npz {x [H,W], wave, J} -> OUT.npz {yl, yh0.. (finest first, (cH,cV,cD) stacked)}"""
    d = np.load(inp)
    J = int(d['J'])
    c = pywt.wavedec2(d['x'], str(d['wave']), 'symmetric', level=J)
    res = {'yl': c[0]}
    for j in range(J):                      # c[1] is the coarsest
        res['yh%d' % (J - 1 - j)] = np.stack(c[1 + j])
    np.savez(out, **res)


if __name__ == '__main__':
    if sys.argv[1] == 'filters':
        filters(sys.argv[2])
    elif sys.argv[1] == 'wavedec2':
        wavedec2(sys.argv[2], sys.argv[3])
    else:
        waverec2(sys.argv[2], sys.argv[3])
