"""TEST INFRASTRUCTURE (not the product).  Pins the one part of the path whose arithmetic lives in a dependency that is absent
from the build image: torchvision's `RandomPerspective`, `RandomErasing` and `functional.affine` behind the reference's
`transforms_fast` (aphantasia/transforms.py:73-83,165-170) and illustrip's `frame_transform` (illustrip.py:130-138).

    python oracle/make_tv_fixture.py            # needs torchvision (any >= 0.8.2, the reference's own requirement) AND /root/reference

It runs the REFERENCE's own `transforms_fast` (imported in place; only kornia / cv2 / imageio / pywt are stubbed, torchvision is
the real one) on seeded cuts and writes tests/golden/tf_fast_224.npz:
    stream_in / stream_out   the reference's transforms_fast applied cut by cut after torch.manual_seed(11); np.random.seed(11):
                             pins the op arithmetic AND the order in which the random parameters are drawn
    persp_*                  T.functional.perspective with explicit start / end points (bilinear, fill 0)
    affine_*                 T.functional.affine with explicit (angle, translate, scale, shear) on a non-square frame
    versions                 torch / torchvision version strings
tests/test_oracle.py::test_torchvision_fixture and tests/test_gpu_kernels.py::test_augment_vs_torchvision_fixture consume the file
when it exists and SKIP (loudly) when it does not.  Nobody has been able to run this script yet: the build image and the GPU boxes
have no torchvision and no network.  Until someone does, SURVEY rows a-8 / f-1 stay "parity unpinned" (DESIGN.md section 2).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden', 'tf_fast_224.npz')
REFERENCE_ROOT = os.environ.get('APH_REFERENCE_ROOT', '/root/reference')

STREAM_SEED, STREAM_CUTS, SIZE = 11, 6, 224
PERSP_CASES = [      # (startpoints, endpoints) as RandomPerspective.get_params returns them for a 224 x 224 cut
    ([[0, 0], [223, 0], [223, 223], [0, 223]], [[12, 30], [200, 7], [215, 190], [25, 219]]),
    ([[0, 0], [223, 0], [223, 223], [0, 223]], [[0, 0], [223, 0], [223, 223], [0, 223]]),          # identity homography
    ([[0, 0], [223, 0], [223, 223], [0, 223]], [[36, 36], [187, 3], [190, 201], [1, 188]]),
]
AFFINE_CASES = [     # (angle, translate [x, y], scale, shear) -- illustrip.py:381-384 style values and the rotation-only form of transforms.py:79
    (0.8, [0.0, 10.0], 1.012, 0.4), (-17.0, [0.0, 0.0], 1.0, 0.0), (30.0, [0.0, 0.0], 1.0, 0.0), (-2.5, [7.0, -3.0], 0.97, -1.2), (0.0, [0.0, 0.0], 1.0, 0.0),
]
FRAME_HW = (72, 120)


def load_reference_transforms():
    """the reference's aphantasia/transforms.py with the REAL torchvision; the other absent third-party modules stubbed"""
    import importlib
    import importlib.machinery
    import torchvision  # noqa: F401  (ImportError here = this script cannot run on this machine)

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, k): return _Any()

    for name in ['kornia', 'kornia.filters', 'kornia.filters.sobel', 'kornia.geometry', 'kornia.geometry.transform', 'cv2', 'imageio', 'pywt', 'pytorch_wavelets']:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
                m.__path__ = []
                m.__getattr__ = lambda k: _Any()
                sys.modules[name] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    for k in [k for k in sys.modules if k == 'aphantasia' or k.startswith('aphantasia.')]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        return importlib.import_module('aphantasia.transforms')
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == 'aphantasia' or k.startswith('aphantasia.')]:
            del sys.modules[k]


def inputs():
    """the seeded inputs both this script and the consuming tests build (never stored twice)"""
    g = torch.Generator().manual_seed(5)
    cuts = torch.rand(STREAM_CUTS, 3, SIZE, SIZE, generator=g)
    frame = torch.rand(1, 3, FRAME_HW[0], FRAME_HW[1], generator=g)
    return cuts, frame


def main():
    import torchvision
    import torchvision.transforms as T
    ref = load_reference_transforms()
    cuts, frame = inputs()
    out = dict(versions=np.array('torch %s, torchvision %s' % (torch.__version__, torchvision.__version__)))
    torch.manual_seed(STREAM_SEED)
    np.random.seed(STREAM_SEED)
    out['stream_out'] = torch.cat([ref.transforms_fast(cuts[c:c + 1]) for c in range(STREAM_CUTS)], 0).numpy()      # utils.py:251: one call per cut
    bil = T.InterpolationMode.BILINEAR
    out['persp_out'] = torch.cat([T.functional.perspective(cuts[i:i + 1], sp, ep, interpolation=bil, fill=0) for i, (sp, ep) in enumerate(PERSP_CASES)], 0).numpy()
    out['rotate_out'] = torch.cat([T.functional.center_crop(T.functional.affine(cuts[i % STREAM_CUTS:i % STREAM_CUTS + 1], a, [0, 0], 1, 0, fill=0, interpolation=bil), [SIZE, SIZE])
                                   for i, (a, _, _, _) in enumerate(AFFINE_CASES)], 0).numpy()                                      # transforms.py:79-81
    out['affine_out'] = torch.cat([T.functional.affine(frame, a, list(t), s, sh, fill=0, interpolation=bil) for (a, t, s, sh) in AFFINE_CASES], 0).numpy()   # illustrip.py:130-138
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, str(out['versions']))


if __name__ == '__main__':
    main()
