"""HBM-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the
same bench.py command ->  profiles/<tag>_pmc_hbm_traffic.{csv,json}.

    python tools/pmc_traffic.py <fetch-dir> <write-dir> <tag> [workload note] [--calib profiles/rNN_pmc_calibration.json]
    (a <tag> ending in `_c4` additionally sums the inverse-DWT kernels per pass: `irdwt_fwd_bytes_per_pass` / `irdwt_bwd_bytes_per_pass`,
     tied to the sha256 of csrc/dwt.hip -- what bench.py --config c4 quotes as roofline.irdwt.traffic)

The number of optimisation steps in the profiled run is DERIVED from the trace -- the launches of `adam_kernel`, one per step -- and
never taken from the command line (round 5 passed a stale "5" while bench.py executed 11 steps and published an irDWT traffic 2.2x too
high).  Both passes must hold the same step count and every per-step family must divide by it, else nothing is written.

Bytes per launch = ff x FETCH_SIZE x 1024 + wf x WRITE_SIZE x 1024 with the factors of tools/pmc/pmc_calib on this part (known 1 GiB
streams per access width, `--calib`; without a calibration file ff = 2 -- FETCH_SIZE reads half of a 16 B/lane stream on gfx950,
MI355X_MICROARCH.md HBM section -- and wf = 1).  The json carries the factors used and the sha256 of the library that was profiled;
bench.py quotes `roofline.traffic` from it only for the same build and flags it stale otherwise."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.environ.get('APH_PMC_OUT', os.path.join(ROOT, 'profiles'))      # (tests write elsewhere)


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under ' + d
    tot, n = defaultdict(float), defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            k = re.sub(r'\(.*', '', row['Kernel_Name'])
            tot[k] += float(row['Counter_Value'])
            n[k] += 1
    return tot, n


def steps_in(n, what):
    """optimisation steps of the profiled run = launches of the Adam kernel (exactly one per step, every variant counted)"""
    s = sum(v for k, v in n.items() if 'adam_kernel' in k)
    if s <= 0:
        raise SystemExit('pmc_traffic: no adam_kernel launch in the %s pass -- not a bench.py / Engine.step trace, refusing to guess the step count' % what)
    return s


def factors(calib):
    """(fetch factor, write factor, per-class table) from a tools/pmc/calib_summary.py json; the 16 B/lane stream is the default class"""
    if not calib:
        return 2.0, 1.0, None
    k = json.load(open(calib))['kernels']
    return k['read_16B_per_lane']['fetch_factor'], k['write_16B_per_lane']['write_factor'], k


# kernels whose dominant streams are NOT 16 B/lane: (substring, read class, write class) -> factors of that class from the calibration
ACCESS_CLASS = (
    ('idwt_', 'read_4B_per_lane', 'write_4B_per_lane'),
    ('crop_resize_strips', 'read_4B_per_lane', 'write_8B_per_lane'),
    ('crop_adjoint', 'read_4B_per_lane', 'write_4B_per_lane'),
)


def main():
    argv = sys.argv[1:]
    calib = None
    if '--calib' in argv:
        i = argv.index('--calib')
        calib = argv[i + 1]
        del argv[i:i + 2]
    fd, wd, tag = argv[0], argv[1], argv[2]
    if tag.isdigit():
        raise SystemExit('pmc_traffic: the step count is no longer an argument (it is derived from the adam_kernel launches); usage: <fetch-dir> <write-dir> <tag> [note]')
    note = argv[3] if len(argv) > 3 else ''
    ft, fn = per_kernel(fd, 'FETCH_SIZE')
    wt, wn = per_kernel(wd, 'WRITE_SIZE')
    steps, steps_w = steps_in(fn, 'FETCH_SIZE'), steps_in(wn, 'WRITE_SIZE')
    if steps != steps_w:
        raise SystemExit('pmc_traffic: the two passes ran different step counts (%d vs %d adam_kernel launches): not the same command, nothing written' % (steps, steps_w))
    ff0, wf0, classes = factors(calib)
    used = {}
    rows = []
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        if not k.startswith(('void aph::', 'aph::', '_ZN3aph')):
            continue
        launches = fn[k]
        if k in wn and wn[k] != launches:
            raise SystemExit('pmc_traffic: %s has %d launches in the FETCH pass and %d in the WRITE pass' % (k[:80], launches, wn[k]))
        ff, wf = ff0, wf0
        if classes:
            for sub, rc, wc in ACCESS_CLASS:
                if sub in k and rc in classes and wc in classes:
                    ff, wf = classes[rc]['fetch_factor'], classes[wc]['write_factor']
                    used[sub] = dict(fetch_factor=ff, write_factor=wf, read_class=rc, write_class=wc)
        fb, wb = ff * 1024 * ft[k] / launches, wf * 1024 * wt.get(k, 0.0) / max(wn.get(k, 1), 1)
        rows.append((k, launches, launches / steps, fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
    gem = [r for r in rows if 'gemm' in r[0] and r[2] >= 1 and 'splitk_reduce' not in r[0]]      # (the ring kernels' signatures contain 'SplitK')
    gl = sum(r[1] for r in gem)
    if gl % steps:
        raise SystemExit('pmc_traffic: %d GEMM launches do not divide by the %d steps of the run: the trace is truncated or mixed, nothing written' % (gl, steps))
    lib = os.path.join(ROOT, 'aphantasia_amd', 'libaphantasia_hip.so')
    sys.path.insert(0, ROOT)
    from bench import gemm_src_sha          # sha256 over the ViT translation unit's sources (bench.py accepts the summary on either hash)
    out = dict(kernel_family='aph::gemm*_f16_kernel (launches occurring every step)', launches=gl,
               traffic_bytes_per_launch=sum(r[5] * 1e6 * r[1] for r in gem) / max(gl, 1),
               per_kernel_MB_per_launch={r[0][-100:]: round(r[5], 2) for r in rows},
               lib_sha256=hashlib.sha256(open(lib, 'rb').read()).hexdigest(), gemm_src_sha256=gemm_src_sha(), workload=note,
               steps_in_run=steps, steps_from='adam_kernel launches (one per optimisation step)', gemm_launches_per_step=gl // steps,
               fetch_factor=ff0, write_factor=wf0, calibration=os.path.relpath(calib, ROOT) if calib else None, access_class_factors=used,
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over the same bench.py command; bytes = fetch_factor x FETCH_SIZE x 1024 + '
                      'write_factor x WRITE_SIZE x 1024, factors from tools/pmc/pmc_calib on known 1 GiB streams (16 B/lane class unless access_class_factors names the kernel)')
    if tag.endswith('_c4'):
        from bench import dwt_src_sha
        lv = sum(r[1] for r in rows if 'idwt_level_kernel' in r[0])
        if lv == 0 or lv % steps:
            raise SystemExit('pmc_traffic: %d idwt_level_kernel launches do not divide by the %d steps of the run, nothing written' % (lv, steps))
        fwd = sum(r[5] * 1e6 * r[1] for r in rows if 'idwt_' in r[0] and 'adjoint' not in r[0]) / steps
        bwd = sum(r[5] * 1e6 * r[1] for r in rows if 'idwt_' in r[0] and 'adjoint' in r[0]) / steps
        out.update(irdwt_fwd_bytes_per_pass=fwd, irdwt_bwd_bytes_per_pass=bwd, dwt_src_sha256=dwt_src_sha(),
                   irdwt_note='sum over idwt_level_kernel + idwt_coarse_kernel (forward) / their adjoints of 2 x FETCH_SIZE + WRITE_SIZE, per optimisation step (one pass each way)')
        print('irDWT: forward %.1f MB / pass, adjoint %.1f MB / pass' % (fwd / 1e6, bwd / 1e6))
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, tag + '_pmc_hbm_traffic.csv'), 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'launches_per_step(%d steps)' % steps, 'fetch_MB_per_launch(calibrated)', 'write_MB_per_launch(calibrated)', 'total_MB_per_launch'])
        w.writerows(rows)
    with open(os.path.join(OUT_DIR, tag + '_pmc_hbm_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for r in rows[:14]:
        print('%-90s %5d launches  fetch %8.1f MB  write %8.1f MB' % (r[0][-90:], r[1], r[3], r[4]))
    print('GEMM family: %.1f MB / launch over %d launches' % (out['traffic_bytes_per_launch'] / 1e6, gl))


if __name__ == '__main__':
    main()
