"""HBM-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the
same bench.py command ->  profiles/<tag>_pmc_hbm_traffic.{csv,json}.

    python tools/pmc_traffic.py <fetch-dir> <write-dir> <steps-in-run> <tag> [workload note]
    (a <tag> ending in `_c4` additionally sums the inverse-DWT kernels per pass: `irdwt_fwd_bytes_per_pass` / `irdwt_bwd_bytes_per_pass`,
     tied to the sha256 of csrc/dwt.hip -- what bench.py --config c4 quotes as roofline.irdwt.traffic)

Bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (FETCH_SIZE reads half of a wide coalesced stream on gfx950:
MI355X_MICROARCH.md, HBM section; WRITE_SIZE as reported, uncalibrated).  The json carries the sha256 of the library that
was profiled; bench.py quotes `roofline.traffic` from it only for the same build and flags it stale otherwise."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def main():
    fd, wd, steps, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ''
    ft, fn = per_kernel(fd, 'FETCH_SIZE')
    wt, wn = per_kernel(wd, 'WRITE_SIZE')
    rows = []
    for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        if not k.startswith(('void aph::', 'aph::', '_ZN3aph')):
            continue
        launches = fn[k]
        fb, wb = 2 * 1024 * ft[k] / launches, 1024 * wt.get(k, 0.0) / max(wn.get(k, 1), 1)
        rows.append((k, launches, launches / steps, fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles', tag + '_pmc_hbm_traffic.csv'), 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'launches_per_step', 'fetch_MB_per_launch(2xFETCH_SIZE)', 'write_MB_per_launch', 'total_MB_per_launch'])
        w.writerows(rows)
    gem = [r for r in rows if 'gemm' in r[0] and r[2] >= 1 and 'splitk_reduce' not in r[0]]      # (the ring kernels' signatures contain 'SplitK')
    gl = sum(r[1] for r in gem)
    lib = os.path.join(ROOT, 'aphantasia_amd', 'libaphantasia_hip.so')
    sys.path.insert(0, ROOT)
    from bench import gemm_src_sha          # sha256 over the ViT translation unit's sources (bench.py accepts the summary on either hash)
    out = dict(kernel_family='aph::gemm*_f16_kernel (launches occurring every step)', launches=gl,
               traffic_bytes_per_launch=sum(r[5] * 1e6 * r[1] for r in gem) / max(gl, 1),
               per_kernel_MB_per_launch={r[0][-100:]: round(r[5], 2) for r in rows},
               lib_sha256=hashlib.sha256(open(lib, 'rb').read()).hexdigest(), gemm_src_sha256=gemm_src_sha(), workload=note,
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over the same bench.py command; FETCH_SIZE doubled '
                      '(gfx950 half-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported')
    if tag.endswith('_c4'):
        from bench import dwt_src_sha
        fwd = sum(r[5] * 1e6 * r[1] for r in rows if 'idwt_' in r[0] and 'adjoint' not in r[0]) / steps
        bwd = sum(r[5] * 1e6 * r[1] for r in rows if 'idwt_' in r[0] and 'adjoint' in r[0]) / steps
        out.update(irdwt_fwd_bytes_per_pass=fwd, irdwt_bwd_bytes_per_pass=bwd, dwt_src_sha256=dwt_src_sha(),
                   irdwt_note='sum over idwt_level_kernel + idwt_coarse_kernel (forward) / their adjoints of 2 x FETCH_SIZE + WRITE_SIZE, per optimisation step (one pass each way)')
        print('irDWT: forward %.1f MB / pass, adjoint %.1f MB / pass' % (fwd / 1e6, bwd / 1e6))
    with open(os.path.join(ROOT, 'profiles', tag + '_pmc_hbm_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for r in rows[:14]:
        print('%-90s %5d launches  fetch %8.1f MB  write %8.1f MB' % (r[0][-90:], r[1], r[3], r[4]))
    print('GEMM family: %.1f MB / launch over %d launches' % (out['traffic_bytes_per_launch'] / 1e6, gl))


if __name__ == '__main__':
    main()
