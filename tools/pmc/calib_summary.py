"""FETCH_SIZE / WRITE_SIZE calibration factors from the two rocprofv3 passes over tools/pmc/pmc_calib (known 1 GiB per kernel):

    python tools/pmc/calib_summary.py <fetch-dir> <write-dir> <out.json>

factor = true bytes / (counter x 1024).  tools/pmc_traffic.py multiplies a kernel's counters by the factor of its access class."""
import csv, glob, json, os, re, sys
csv.field_size_limit(1 << 30)
BYTES = float(1 << 30)


def per_kernel(d, counter):
    tot, n = {}, {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            k = row['Kernel_Name']
            tot[k] = tot.get(k, 0.0) + float(row['Counter_Value'])
            n[k] = n.get(k, 0) + 1
    return {k: tot[k] / n[k] for k in tot}


def label(k):
    m = re.search(r'(rd_stride|rd|wr)_kernel(?:<(.*)>)?\s*\(', k) or re.search(r'(rd_stride|rd|wr)_kernel(?:<(.*)>)?', k)
    if not m:
        return None
    arg = (m.group(2) or '').strip()
    t = '2B' if 'short' in arg else '8B' if ('float2' in arg or ', 2u' in arg) else '16B' if ('float4' in arg or ', 4u' in arg) else '4B' if 'float' in arg else arg
    return {'rd': 'read_%s_per_lane' % t, 'wr': 'write_%s_per_lane' % t, 'rd_stride': 'read_4B_per_lane_stride2'}[m.group(1)]


def main():
    fd, wd, out = sys.argv[1:4]
    f, w = per_kernel(fd, 'FETCH_SIZE'), per_kernel(wd, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        lb = label(k)
        if lb is None:
            continue
        e = res.setdefault(lb, {})
        if k in f:
            e['FETCH_SIZE_KB'] = f[k]
            if lb.startswith('read') and f[k] > 0:
                e['fetch_factor'] = BYTES / (f[k] * 1024)
        if k in w:
            e['WRITE_SIZE_KB'] = w[k]
            if lb.startswith('write') and w[k] > 0:
                e['write_factor'] = BYTES / (w[k] * 1024)
    j = dict(true_bytes_per_kernel=BYTES, note='factor = true bytes / (counter x 1024); 1 GiB streams (4x the Infinity Cache), grid 2048 x 256 threads; '
             'stride-2 read touches every 128-byte line of the GiB and uses half of each (its factor is against the LINES touched = 1 GiB)', kernels=res)
    json.dump(j, open(out, 'w'), indent=1)
    for k, v in res.items():
        print('%-28s %s' % (k, {a: round(b, 4) for a, b in v.items()}))


if __name__ == '__main__':
    main()
