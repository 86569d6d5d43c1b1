"""Per-kernel sums of every counter found in rocprofv3 --pmc output directories:  python tools/pmc_table.py <csv-out> <dir> [<dir> ...]
(one row per kernel, launches and each counter summed over launches; ratios to SQ_WAVE_CYCLES appended when present)"""
import csv, glob, os, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
out, dirs = sys.argv[1], sys.argv[2:]
tot, launches, names = defaultdict(lambda: defaultdict(float)), defaultdict(set), []
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name'])
            if not k.startswith(('void aph::', 'aph::', '_ZN3aph')):
                continue
            k = k[-70:]
            c = r['Counter_Name']
            if c not in names:
                names.append(c)
            tot[k][c] += float(r['Counter_Value'])
            launches[k].add((d, r['Dispatch_Id']))
rows = []
for k in sorted(tot, key=lambda k: -tot[k].get('SQ_WAVE_CYCLES', tot[k].get('GRBM_GUI_ACTIVE', 0))):
    n = max(1, len(launches[k]) // max(1, len(dirs)))
    rows.append([k, n] + [tot[k].get(c, '') for c in names])
with open(out, 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'launches'] + names)
    w.writerows(rows)
want = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INST_CYCLES_VMEM_RD',
        'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TA_TA_BUSY_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum']
have = [c for c in want if c in names]
print('%-60s %6s ' % ('kernel', 'n') + ' '.join('%12s' % c.replace('SQ_', '').replace('_sum', '')[:12] for c in have))
for r in rows[:24]:
    d = dict(zip(names, r[2:]))
    wc = d.get('SQ_WAVE_CYCLES') or 0
    cells = []
    for c in have:
        v = d.get(c, '')
        if v == '':
            cells.append('%12s' % '-')
        elif c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES' and not c.startswith('SQ_INSTS') and wc:
            cells.append('%12.3f' % (v / wc))
        else:
            cells.append('%12.3g' % v)
    print('%-60s %6d ' % (r[0][-60:], r[1]) + ' '.join(cells))
