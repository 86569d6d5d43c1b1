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
    # the drop-in boundary + the test / measurement hooks (separate header: not part of the boundary)
    hdr = open(os.path.join(ROOT, 'include', 'aphantasia_hip.h')).read() + open(os.path.join(ROOT, 'include', 'aphantasia_hip_test.h')).read()
    declared = set(re.findall(r'\b(aph_[a-z0-9_]+)\s*\(', hdr))
    boundary = set(re.findall(r'\b(aph_[a-z0-9_]+)\s*\(', open(os.path.join(ROOT, 'include', 'aphantasia_hip.h')).read()))
    assert not ({'aph_gemm_f16', 'aph_gemm_f16_ld', 'aph_vit_profile', 'aph_vit_profile_read'} & boundary)
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for name in declared:
        assert hasattr(product_lib.cdll, name), name          # dlsym of every header symbol
    assert product_lib.cdll.aph_version() >= 200
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
    import depth.depth as dd                                   # illustrip.py:30 `from depth import depth`
    for n in ('grid_warp', 'depthwarp', 'resize', 'InferDepthAny'):
        assert callable(getattr(dd, n))
    with pytest.raises(RuntimeError, match='Depth-Anything'):
        dd.InferDepthAny('b', path=None)                       # no checkpoint: refuse, do not invent a depth map


def test_resume_from_image_helpers_match_reference_golden(tmp_path):
    """inv_sigmoid / un_rgb / un_spectrum / img2fft (image.py:179-220) vs outputs of the reference's own functions
    (tests/golden/resume_img.npz, generated by oracle/make_goldens.py through the stub-import shim)"""
    import numpy as np
    import torch
    from aphantasia_amd import image as I
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resume_img.npz'))
    x = torch.from_numpy(g['x'])
    assert torch.equal(I.inv_sigmoid(x), torch.from_numpy(g['inv_sigmoid']))
    np.testing.assert_allclose(I.un_rgb(g['img'], colors=1.5).numpy(), g['un_rgb_c15'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(I.un_rgb(x, colors=1.0).numpy(), g['un_rgb_tensor'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(I.un_spectrum(torch.from_numpy(g['spec']), 1.5).numpy(), g['un_spectrum'], rtol=1e-6, atol=1e-9)
    want = g['img2fft']
    got = I.img2fft(g['img'], 1.5, 1.5).numpy()
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()


def test_zero_effective_samples_is_reported():
    """C1 of the survey: `--samples 1` with -tf fast derates to 0 cuts (upstream crashes in torch.cat([]))"""
    import clip_fft
    a = clip_fft.get_args(['-t', 'x', '--samples', '1'])
    a.modsize = 224
    assert clip_fft.derate_samples(a) == 0
    with pytest.raises(SystemExit):
        clip_fft.check_samples(0)
    clip_fft.check_samples(1)


@pytest.mark.parametrize('align', ['uniform', 'central', 'overscan'])
def test_bulk_draws_have_the_reference_distributions(align):
    """Engine(rng='bulk') (the default) draws the crop / augment tables vectorised from a numpy Generator; they must follow the
    same distributions and bounds as the reference-order draws (utils.py:222-251, transforms.py:165-170)"""
    import numpy as np
    import torch
    from aphantasia_amd.utils import draw_crop_params, draw_crop_params_bulk
    from aphantasia_amd import transforms
    from aphantasia_amd.transforms import pack_aug
    h, w, size, n = 720, 1280, 224, 4000
    torch.manual_seed(1); np.random.seed(1)
    ta, aa = draw_crop_params(n, size, h, w, align, 0.4, transforms.transforms_fast)
    tb, ab = draw_crop_params_bulk(n, size, h, w, align, 0.4, transforms.transforms_fast, np.random.default_rng(2))
    aa = pack_aug(aa).numpy()
    ph, pw = (h, w) if align != 'overscan' else (int(1.5 * h), int(1.5 * w))
    for t in (ta, tb):
        assert t[:, 0].min() >= size and t[:, 0].max() <= min(h, w)
        assert (t[:, 1] >= 0).all() and (t[:, 1] + t[:, 0] <= pw).all() and (t[:, 2] >= 0).all() and (t[:, 2] + t[:, 0] <= ph).all()
    for col in range(3):                                      # means / spreads of csize, offx, offy agree within sampling error
        a, b = ta[:, col].astype(np.float64), tb[:, col].astype(np.float64)
        assert abs(a.mean() - b.mean()) < 5 * a.std() / np.sqrt(n) + 1e-9, (col, a.mean(), b.mean())
        assert abs(a.std() - b.std()) < 0.08 * a.std() + 1e-9, (col, a.std(), b.std())
    big = 0.9 * min(h, w)
    assert abs((ta[:, 0] >= big).mean() - (tb[:, 0] >= big).mean()) < 0.04                  # macro share (~0.4 + tail)
    for flag_col, p in ((8, 0.2), (15, 59.0 / 80.0)):        # perspective probability; rotation stage on for the non-zero angles (59 of 80 choices)
        assert abs((aa[:, flag_col] != 0).mean() - p) < 0.03 and abs((ab[:, flag_col] != 0).mean() - p) < 0.03
    assert abs((aa[:, 11] > 0).mean() - (ab[:, 11] > 0).mean()) < 0.04                      # erase probability
    za, zb = (np.abs(aa[:, 14]) < 1e-7).mean(), (np.abs(ab[:, 14]) < 1e-7).mean()           # share of angle 0 (21 of 80 choices)
    assert abs(za - 21 / 80) < 0.03 and abs(zb - 21 / 80) < 0.03
    ma, mb = aa[:, 0:8][aa[:, 8] != 0].mean(0), ab[:, 0:8][ab[:, 8] != 0].mean(0)             # perspective coefficients
    assert np.abs(ma - mb)[[0, 1, 3, 4]].max() < 0.05 and np.abs(ma - mb)[[2, 5]].max() < 8.0 and np.abs(ma - mb)[[6, 7]].max() < 1e-3, (ma, mb)


def test_error_convention_on_bad_arguments(product_lib):
    """every entry point returns int (< 0 on error) and leaves a message for aph_last_error(); argument
    checks come before any device work, so this runs without a GPU"""
    import ctypes
    L = product_lib.cdll
    null = None
    cases = {
        'aph_synth_fft_fwd': lambda: L.aph_synth_fft_fwd(null, null, null, null, ctypes.c_float(1.0), null, 1, null, null, null),
        'aph_sample_fwd': lambda: L.aph_sample_fwd(null, null, null, null, null, null, 2, null),
        'aph_sample_bwd': lambda: L.aph_sample_bwd(null, null, ctypes.c_float(1.0), null, null, null, null, 2, null),
        'aph_vit_forward': lambda: L.aph_vit_forward(null, null, 1, null, null),
        'aph_vit_backward': lambda: L.aph_vit_backward(null, null, 1, null, ctypes.c_float(1.0), null),
        'aph_vit_create': lambda: L.aph_vit_create(224, 32, 700, 12, 12, 512, 8, null),
        'aph_sim_loss': lambda: L.aph_sim_loss(null, 1, 1, null, null, null, 1, 1, 1, 0, 1, ctypes.c_float(1.0), ctypes.c_float(1.0), null, null, null, null),
        'aph_adam_step': lambda: L.aph_adam_step(null, null, null, null, null, null, 0, ctypes.c_size_t(1), null),
        'aph_adam_step_guarded': lambda: L.aph_adam_step_guarded(null, null, null, null, null, null, 0, ctypes.c_size_t(1), null, null),
        'aph_idwt_level_fwd': lambda: L.aph_idwt_level_fwd(null, 1, 1, null, 1, 1, 3, null, null, 6, ctypes.c_float(1.0), null, null),
        'aph_rgb_priors': lambda: L.aph_rgb_priors(null, 4, 4, ctypes.c_float(0.45), ctypes.c_float(0.17), ctypes.c_float(1.0), null, null, null, null),
        'aph_rgb_sharp': lambda: L.aph_rgb_sharp(null, 4, 4, ctypes.c_float(1.0), null, null, null, null),
        'aph_frame_affine': lambda: L.aph_frame_affine(null, 3, 4, 4, null, null, null),
        'aph_gemm_f16': lambda: L.aph_gemm_f16(null, null, 1, 128, 64, null, null),
        'aph_triangle_blur': lambda: L.aph_triangle_blur(null, 3, 8, 8, 5, ctypes.c_float(2.0), ctypes.c_float(0.5), null, null),
        'aph_resize_bicubic': lambda: L.aph_resize_bicubic(null, 3, 8, 8, null, 4, 4, null),
        'aph_flip_w': lambda: L.aph_flip_w(null, null, 3, 8, 8, null, null),
        'aph_grid_warp': lambda: L.aph_grid_warp(null, null, 3, 8, 8, ctypes.c_float(0.3), ctypes.c_float(0.0), ctypes.c_float(0.0), ctypes.c_float(0.5),
                                                 ctypes.c_float(0.05), null, null, null),
        'aph_attn_test': lambda: L.aph_attn_test(null, null, null, null, null, null, 1, 50, 12, 0, null),
        'aph_allreduce_f32': lambda: L.aph_allreduce_f32(null, null, ctypes.c_size_t(1), null),
    }
    for name, call in cases.items():
        rc = call()
        assert rc < 0, name
        assert product_lib.last_error(), name


def _uid_worker(rank, world, port, mode, q):
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from aphantasia_amd import comm as acomm
    uid = bytes(range(128)) if rank == 0 else None
    if mode == 'file':
        got = acomm._file_exchange(rank, world, uid, 'test_%d' % port, timeout=60.0)
    else:
        got = acomm._gloo_exchange(rank, world, uid)
    q.put((rank, got))


@pytest.mark.parametrize('mode', ['file', 'gloo'])
def test_rccl_unique_id_exchange_two_processes(mode):
    """the two out-of-band paths that carry the 128-byte RCCL id when the caller has no torch.distributed group
    (comm.create: clip_fft.py --ranks -> rendezvous file keyed by the launch; torchrun + clip_fft.py -> throw-away gloo group)"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + (7 if mode == 'gloo' else 0)
    procs = [ctx.Process(target=_uid_worker, args=(r, 2, port, mode, q)) for r in (1, 0)]        # the reader starts first
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res[0] == bytes(range(128)) and res[1] == bytes(range(128))


def test_bench_traffic_summary_matching(tmp_path, monkeypatch):
    """bench.py quotes `roofline.traffic` from a committed PMC summary only when that summary was taken on the library being timed or on
    byte-identical GEMM sources; otherwise it is flagged stale.  The newest matching file wins, file order is by name (not mtime)."""
    import json
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'lib_sha', lambda: 'L' * 64)
    monkeypatch.setattr(bench, 'gemm_src_sha', lambda: 'G' * 64)

    def write(name, **kw):
        (prof / name).write_text(json.dumps(dict(traffic_bytes_per_launch=float(len(name)), **kw)))
    assert bench.pmc_traffic('r*_pmc_hbm_traffic*.json') == (None, None, None, None)
    write('r01_pmc_hbm_traffic.json', lib_sha256='x')
    assert bench.pmc_traffic('r*_pmc_hbm_traffic*.json')[2:] == (True, None)                 # nothing matches: the newest one, flagged stale
    write('r02_pmc_hbm_traffic.json', lib_sha256='y', gemm_src_sha256='G' * 64)
    write('r03_pmc_hbm_traffic.json', lib_sha256='z')
    t, src, stale, match = bench.pmc_traffic('r*_pmc_hbm_traffic*.json')
    assert (os.path.basename(src), stale, match) == ('r02_pmc_hbm_traffic.json', False, 'gemm_sources')
    write('r04_pmc_hbm_traffic.json', lib_sha256='L' * 64)
    t, src, stale, match = bench.pmc_traffic('r*_pmc_hbm_traffic*.json')
    assert (os.path.basename(src), stale, match) == ('r04_pmc_hbm_traffic.json', False, 'library')


def test_committed_traffic_summaries_match_sources():
    """RELEASE CHECK: the newest committed PMC summaries (by name) were taken on this checkout's sources -- the GEMM-family one on the ViT
    translation unit, the C4 one on csrc/dwt.hip -- so that bench.py quotes `roofline.traffic` / `irdwt.traffic` un-flagged; and both carry
    the step count derived from the trace (round 6: tools/pmc_traffic.py no longer takes it from argv)"""
    import glob
    import json
    import bench
    c2 = sorted(p for p in glob.glob(os.path.join(bench.ROOT, 'profiles', 'r[0-9][0-9]_pmc_hbm_traffic.json')))[-1]
    c4 = sorted(glob.glob(os.path.join(bench.ROOT, 'profiles', 'r[0-9][0-9]_c4_pmc_hbm_traffic.json')))[-1]
    j2, j4 = json.load(open(c2)), json.load(open(c4))
    assert j2['gemm_src_sha256'] == bench.gemm_src_sha(), '%s was taken on other GEMM sources: re-run `bash tools/gpu.sh rNN pmc`' % os.path.basename(c2)
    assert j4['dwt_src_sha256'] == bench.dwt_src_sha(), '%s was taken on another csrc/dwt.hip' % os.path.basename(c4)
    for j in (j2, j4):
        assert j['steps_in_run'] >= 1 and j['steps_from'].startswith('adam_kernel')
    # the irDWT moves its algorithmic bytes (266.4 MB per pass at 3840x2160 db3) with modest overhead -- not the 3x the stale step count of round 5 reported
    assert 2.6e8 < j4['irdwt_fwd_bytes_per_pass'] < 1.6 * 2.664e8 and 2.6e8 < j4['irdwt_bwd_bytes_per_pass'] < 1.6 * 2.664e8


def _write_counters(d, counter, launches):
    """a minimal rocprofv3 counter_collection.csv: `launches` = [(kernel name, value), ...]"""
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'x_counter_collection.csv'), 'w') as f:
        f.write('Kernel_Name,Counter_Name,Counter_Value\n')
        for k, v in launches:
            f.write('"%s",%s,%f\n' % (k, counter, v))


def test_pmc_traffic_derives_the_step_count_from_the_trace(tmp_path, monkeypatch):
    """tools/pmc_traffic.py: the number of optimisation steps of the profiled run comes from the adam_kernel launches of the trace, never
    from the command line (round 5 published an irDWT traffic 2.2x too high through a stale argv step count); passes of different length,
    traces without an Adam launch and family counts that do not divide by the step count are refused and nothing is written"""
    import importlib.util
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'pmc_traffic.py')
    gemm, adam, lvl, adj = 'void aph::gemm_ws_kernel<A>(x)', 'void aph::adam_kernel<0>(y)', 'void aph::idwt_level_kernel<3>(z)', 'void aph::idwt_level_adjoint_kernel<3>(z)'
    steps = 11

    def trace(n_steps, n_gemm=4, n_lvl=5):
        rows = []
        for _ in range(n_steps):
            rows += [(gemm, 1000.0)] * n_gemm + [(lvl, 500.0)] * n_lvl + [(adj, 250.0)] * n_lvl + [(adam, 10.0)]
        return rows
    fd, wd, out = str(tmp_path / 'f'), str(tmp_path / 'w'), str(tmp_path / 'out')
    _write_counters(fd, 'FETCH_SIZE', trace(steps))
    _write_counters(wd, 'WRITE_SIZE', trace(steps))
    env = dict(os.environ, APH_PMC_OUT=out)
    r = subprocess.run([sys.executable, tool, fd, wd, 'rXX_c4', 'note'], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.load(open(os.path.join(out, 'rXX_c4_pmc_hbm_traffic.json')))
    assert j['steps_in_run'] == steps and j['gemm_launches_per_step'] == 4
    assert abs(j['traffic_bytes_per_launch'] - (2 * 1000 + 1000) * 1024) < 1
    assert abs(j['irdwt_fwd_bytes_per_pass'] - 5 * (2 * 500 + 500) * 1024) < 1 and abs(j['irdwt_bwd_bytes_per_pass'] - 5 * (2 * 250 + 250) * 1024) < 1
    # the old calling convention (a step count as the third argument) is refused loudly
    r = subprocess.run([sys.executable, tool, fd, wd, '5', 'rYY'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'derived' in (r.stderr + r.stdout) and not os.path.exists(os.path.join(out, 'rYY_pmc_hbm_traffic.json'))
    # passes of different length
    _write_counters(wd, 'WRITE_SIZE', trace(steps - 1))
    r = subprocess.run([sys.executable, tool, fd, wd, 'rZZ'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'different step counts' in (r.stderr + r.stdout) and not os.path.exists(os.path.join(out, 'rZZ_pmc_hbm_traffic.json'))
    # a truncated trace: GEMM launches that do not divide by the steps
    _write_counters(fd, 'FETCH_SIZE', trace(steps) + [(gemm, 1.0)])
    _write_counters(wd, 'WRITE_SIZE', trace(steps) + [(gemm, 1.0)])
    r = subprocess.run([sys.executable, tool, fd, wd, 'rZZ'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'do not divide' in (r.stderr + r.stdout) and not os.path.exists(os.path.join(out, 'rZZ_pmc_hbm_traffic.json'))
    # no Adam launch at all
    _write_counters(fd, 'FETCH_SIZE', [(gemm, 1.0)])
    r = subprocess.run([sys.executable, tool, fd, wd, 'rZZ'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'adam_kernel' in (r.stderr + r.stdout)


_LADDER_SUP = '''
import sys, json
sys.path.insert(0, %r)
from aphantasia_amd.comm import ladder
rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def mk(k, name, port):
    code = {'hang': "import time\\nif %%d == 1: time.sleep(1000)\\nprint('hang-rung')" %% rank,
            'crash': "import sys\\nsys.exit(3 if %%d == 0 else 0)" %% rank,
            'good': "print('{\\"port\\": %%d, \\"rank\\": %%d}')" %% (port, rank)}[name]
    return [sys.executable, '-c', code], {'X_RUNG': name}
k, rec, out = ladder(rank, world, sys.argv[4].split(','), mk, d, budget_s=3.0, grace_s=4.0)
print(json.dumps(dict(k=k, rec=rec, out=open(out).read() if out else None)))
'''


@pytest.mark.parametrize('rungs,want_k', [('hang,crash,good', 2), ('hang,crash', None)])
def test_multi_rank_ladder_survives_hangs_and_crashes(tmp_path, rungs, want_k):
    """aphantasia_amd.comm.ladder (what bench.py --gpus N runs its ranks under): a worker that HANGS on one rank is killed at the rung's
    budget, a worker that exits non-zero on one rank fails the rung on every rank (all or none), and every supervisor arrives at the same
    rung -- or, when every rung fails, at None with the full record (bench.py then still prints a line)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ps = [subprocess.Popen([sys.executable, '-c', _LADDER_SUP % root, str(r), '2', str(tmp_path / 'sup'), rungs], stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in ps]
    assert outs[0]['k'] == outs[1]['k'] == want_k
    strip = lambda rec: [(r['rung'], r['status']) for r in rec]
    assert strip(outs[0]['rec']) == strip(outs[1]['rec'])
    rec = outs[0]['rec']
    assert rec[0]['rung'] == 'hang' and rec[0]['status'][0] == 'ok' and 'killed' in rec[0]['status'][1]
    assert rec[1]['rung'] == 'crash' and 'exit code 3' in rec[1]['status'][0] and rec[1]['status'][1] == 'ok'
    if want_k is not None:
        assert rec[2]['status'] == ['ok', 'ok']
        a, b = json.loads(outs[0]['out']), json.loads(outs[1]['out'])
        assert a['port'] == b['port'] and (a['rank'], b['rank']) == (0, 1)         # both workers of a rung get rank 0's port
    else:
        assert outs[0]['out'] is None


def test_pad_up_to_and_tile_pad_semantics():
    """aphantasia_amd.utils.pad_up_to / tile_pad (the API-compatibility helpers; the fused sampler wraps coordinates instead): periodic
    'centr' extension equals the oracle's restatement of utils.py:152-190 (itself pinned to the reference through the overscan / overmax
    goldens), 'side' appends after the image, 'symm' mirrors with the edge sample repeated, the gather is differentiable"""
    from aphantasia_amd import utils as U
    from oracle import reference_path as R
    torch.manual_seed(0)
    for (h, w, H, W) in [(5, 7, 9, 12), (4, 4, 4, 9), (6, 3, 15, 3), (3, 5, 20, 31), (48, 80, 72, 120)]:
        x = torch.randn(2, 3, h, w)
        assert torch.equal(U.pad_up_to(x, (H, W)), R.pad_up_to(x, (H, W)))
        side = U.pad_up_to(x, (H, W), 'side')
        assert torch.equal(side[..., :h, :w], x)
        assert torch.equal(side, torch.cat([torch.cat([x] * (H // h + 1), 2)[..., :H, :]] * (W // w + 1), 3)[..., :W])
    x = torch.arange(12.).reshape(1, 1, 3, 4)
    m = U.tile_pad(x, (2, 5, 1, 4), symm=True)
    assert m.shape == (1, 1, 8, 11)
    assert m[0, 0, 1].tolist() == [1, 0, 0, 1, 2, 3, 3, 2, 1, 0, 0]              # row 0 mirrored: edge sample repeated, period 2w
    assert m[0, 0, :, 2].tolist() == [0, 0, 4, 8, 8, 4, 0, 0]
    assert torch.equal(U.pad_up_to(x, (3, 4)), x)
    xg = torch.randn(1, 2, 3, 4, requires_grad=True)
    U.pad_up_to(xg, (6, 8)).sum().backward()
    assert torch.equal(xg.grad, torch.full_like(xg, 4.0))
