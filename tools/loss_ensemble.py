"""The HIP path's free-running stress-weight loss curves against the ENSEMBLE of fp32 CPU oracle trajectories under
tests/golden/ensemble/ (oracle/make_loss_ensemble.py: 8 weight seeds x {32, 48, 95} cuts + 2 more crop seeds, 60 steps each), in
both precision modes and under several summation ORDERS of the small GEMMs, so that the default mode is chosen -- and its 1e-3
asserted -- on a distribution, not on one trajectory (VERDICT r5 item 1 / ADVICE r5).

    python tools/loss_ensemble.py [out-prefix]          # writes <prefix>.json and <prefix>.txt (default gpurun_out/precision_ensemble)

Per (member, mode, order): max |d loss| over the 60 steps, the number of steps past 1e-3, the first such step, |d loss| at the
last step.  Modes: `f16` = f16 MFMA operands everywhere (the reference's own GPU dtype, clip_fft.py:119), `split` = the
split-precision forward (Engine(precise=True)).  Orders -- perturbations of the ROUNDING SEQUENCE that leave the single-step errors
where they are: `rs1` the shipped kernel routing, `rs2` the register-staged split-K kernel for every small-M GEMM (another summation
order of the class-row GEMMs), `ls2048` half the loss scale of the f16 backward (every backward rounding lands elsewhere).  (`rs0` of
the first run was bit-identical to `rs1` at these batch sizes -- the routing it switches only applies to one- or two-cut batches -- and
was dropped.)  Extra variants of the f16 mode: `gradf16` = the patch gradient handed to the sampler adjoint as f16 instead of f32
(Engine(grad_f16=True)); `stream16` = the ViT backward's residual-stream gradient kept in f16 only
(aph_vit_set_grad_stream_f16: VERDICT r5 item 7, LayerNorm-backward traffic 117 -> 73 MB per launch).
"""
import glob, json, os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphantasia_amd import clip as aclip, transforms, _ffi
from aphantasia_amd.engine import Engine
from aphantasia_amd.weights import stress_visual_weights, visual_config
from oracle import reference_path as R

H, W = 720, 1280
ORDERS = (('rs1', dict(rs=1)), ('rs2', dict(rs=2)), ('ls2048', dict(rs=1, loss_scale=2048.0)))
EXTRA = (('f16+gradf16', dict(precise=False, rs=1, grad_f16=True)), ('f16+stream16', dict(precise=False, rs=1, stream16=1)))      # (mode label, settings): reported beside the two modes, order 'rs1'


def seed_all(s):
    torch.manual_seed(s); np.random.seed(s)


def members():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'ensemble', 'stress_w*_c*_s*.npz'))):
        ws, cs, S = (int(t[1:]) for t in os.path.basename(f)[:-4].split('_')[1:])
        out.append((ws, cs, S, f))
    return out


def run_member(model, S, cs, want, precise, steps=None, **kw):
    seed_all(0)
    p0 = R.fft_params_init([1, 3, H, W])
    target = torch.randn(1, 512, generator=torch.Generator().manual_seed(2))
    eng = Engine(p0.cuda().contiguous(), H, W, model, S, [(target, -1.0)], sim='mix', transform=transforms.normalize(), rng='reference', precise=precise, **kw)
    seed_all(cs)
    n = len(want) if steps is None else steps
    got = np.zeros(n)
    for i in range(n):
        got[i] = float(eng.step(R.draw_crop_table(S, 224, H, W, 'uniform', 0.4)))
    skipped = int(eng.guard[0])
    del eng
    return got, skipped


def main(prefix):
    L = _ffi.lib()
    cfg = visual_config('ViT-B/32')
    rows = []
    by_ws = {}
    for ws, cs, S, f in members():
        by_ws.setdefault((ws, S), []).append((cs, f))
    for (ws, S), lst in sorted(by_ws.items()):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = aclip.CLIPModel('ViT-B/32', cfg, stress_visual_weights(cfg, ws), None, S)
        for cs, f in lst:
            want = np.load(f)['loss']
            runs = [(mode, oname, dict(o, precise=precise)) for oname, o in ORDERS for mode, precise in (('f16', False), ('split', True))]
            runs += [(mode, 'rs1', dict(o)) for mode, o in EXTRA]
            for mode, oname, o in runs:
                o = dict(o)
                prev = L.cdll.aph_gemm_set_rs(o.pop('rs'))
                prev16 = L.cdll.aph_vit_set_grad_stream_f16(o.pop('stream16', 0))
                try:
                    got, skipped = run_member(model, S, cs, want, o.pop('precise'), **o)
                finally:
                    L.cdll.aph_gemm_set_rs(prev)
                    L.cdll.aph_vit_set_grad_stream_f16(prev16)
                d = np.abs(got - want)
                over = np.nonzero(d > 1e-3)[0]
                rows.append(dict(weights=ws, crops=cs, cuts=S, mode=mode, order=oname, max=float(d.max()), argmax=int(d.argmax()),
                                 n_over=int(len(over)), first_over=int(over[0]) if len(over) else None, last=float(d[-1]),
                                 mean=float(d.mean()), finite=bool(np.isfinite(got).all()), skipped=skipped))
                print('w%d c%d s%-3d %-11s %-6s  max %.2e (step %2d)  steps past 1e-3: %2d  mean %.2e' % (ws, cs, S, mode, oname, d.max(), d.argmax(), len(over), d.mean()), flush=True)
        del model
        torch.cuda.empty_cache()
    summary = summarise(rows)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    import hashlib
    json.dump(dict(lib_sha256=hashlib.sha256(open(_ffi.LIB_PATH, 'rb').read()).hexdigest()[:16], device=torch.cuda.get_device_name(0), rows=rows, summary=summary),
              open(prefix + '.json', 'w'), indent=1)
    with open(prefix + '.txt', 'w') as f:
        f.write(table(rows, summary))
    print(table(rows, summary))


def summarise(rows):
    out = {}
    modes = ['f16', 'split'] + [m for m, _ in EXTRA if any(r['mode'] == m for r in rows)]
    for mode in modes:
        for order in [o for o, _ in ORDERS] + ['any']:
            sel = [r for r in rows if r['mode'] == mode and (order == 'any' or r['order'] == order)]
            if not sel:
                continue
            mx = np.array([r['max'] for r in sel])
            out['%s/%s' % (mode, order)] = dict(n=len(sel), exceed_1e3=int((mx > 1e-3).sum()), median_max=float(np.median(mx)), p90_max=float(np.quantile(mx, 0.9)),
                                                worst=float(mx.max()), mean_of_mean=float(np.mean([r['mean'] for r in sel])))
        for S in sorted({r['cuts'] for r in rows}):
            sel = [r for r in rows if r['mode'] == mode and r['cuts'] == S]
            mx = np.array([r['max'] for r in sel])
            out['%s/s%d' % (mode, S)] = dict(n=len(sel), exceed_1e3=int((mx > 1e-3).sum()), median_max=float(np.median(mx)), worst=float(mx.max()))
    # paired comparison on the same (member, order): how often is split closer than f16, and a sign-test p-value
    key = lambda r: (r['weights'], r['crops'], r['cuts'], r['order'])
    f16 = {key(r): r['max'] for r in rows if r['mode'] == 'f16'}
    sp = {key(r): r['max'] for r in rows if r['mode'] == 'split'}
    ks = sorted(set(f16) & set(sp))
    wins = sum(sp[k] < f16[k] for k in ks)
    from math import comb
    n = len(ks)
    p = sum(comb(n, i) for i in range(0, min(wins, n - wins) + 1)) * 2 / 2 ** n if n else None
    out['paired'] = dict(n=n, split_closer=wins, f16_closer=n - wins, sign_test_p=min(1.0, p) if p is not None else None,
                         median_ratio_split_over_f16=float(np.median([sp[k] / f16[k] for k in ks])) if ks else None)
    # exceedance counts of the two modes over the same cells (Fisher's exact test, two-sided): does split reduce the RATE of curves past 1e-3?
    a, b = sum(f16[k] > 1e-3 for k in ks), sum(sp[k] > 1e-3 for k in ks)
    if n:
        from scipy.stats import fisher_exact
        out['exceedance'] = dict(n=n, f16_past_1e3=int(a), split_past_1e3=int(b), fisher_exact_p=float(fisher_exact([[a, n - a], [b, n - b]])[1]))
    for m, _ in EXTRA:        # an extra variant against its base mode (f16, shipped order), paired per member
        base = {(r['weights'], r['crops'], r['cuts']): r['max'] for r in rows if r['mode'] == 'f16' and r['order'] == 'rs1'}
        var = {(r['weights'], r['crops'], r['cuts']): r['max'] for r in rows if r['mode'] == m}
        kk = sorted(set(base) & set(var))
        if kk:
            out['paired_%s_vs_f16' % m] = dict(n=len(kk), variant_closer=int(sum(var[k] < base[k] for k in kk)), median_ratio=float(np.median([var[k] / base[k] for k in kk])),
                                               variant_past_1e3=int(sum(var[k] > 1e-3 for k in kk)), base_past_1e3=int(sum(base[k] > 1e-3 for k in kk)))
    # spread across orders of one (member, mode): max / min of the max-|d loss|
    spread = {}
    for mode in ('f16', 'split'):
        g = {}
        for r in rows:
            if r['mode'] == mode:
                g.setdefault((r['weights'], r['crops'], r['cuts']), []).append(r['max'])
        ratios = [max(v) / min(v) for v in g.values() if len(v) > 1]
        spread[mode] = dict(median=float(np.median(ratios)), worst=float(np.max(ratios))) if ratios else None
    out['order_spread_max_over_min'] = spread
    return out


def table(rows, summary):
    lines = ['# tools/loss_ensemble.py: max |d loss| over 60 free-running steps against the fp32 CPU oracle ensemble (stress weights, 1280x720, -tf none)',
             '# columns: weights-seed crops-seed cuts | per order (rs1 = shipped routing, rs2, ls2048): f16 / split ; then the extra variants (order rs1) ; * = past 1e-3', '']
    keys = sorted({(r['weights'], r['crops'], r['cuts']) for r in rows}, key=lambda k: (k[2], k[1], k[0]))
    cell = {(r['weights'], r['crops'], r['cuts'], r['order'], r['mode']): r for r in rows}
    lines.append('%-14s' % 'member' + ''.join('   %-23s' % (o + ' f16 / split') for o, _ in ORDERS) + ''.join('   %-12s' % m for m, _ in EXTRA))
    for k in keys:
        s = 'w%d c%-2d s%-3d   ' % k
        for o, _ in ORDERS:
            a, b = cell.get(k + (o, 'f16')), cell.get(k + (o, 'split'))
            fmt = lambda r: ('%.2e%s' % (r['max'], '*' if r['max'] > 1e-3 else ' ')) if r else '   --    '
            s += '   %s / %s ' % (fmt(a), fmt(b))
        for m, _ in EXTRA:
            s += '   %s   ' % fmt(cell.get(k + ('rs1', m)))
        lines.append(s)
    lines.append('')
    for k, v in summary.items():
        lines.append('%-28s %s' % (k, json.dumps(v)))
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'precision_ensemble'))
