"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite): python tools/prof_summary.py <dir> <steps-in-run> [csv-out]"""
import csv, glob, re, sqlite3, sys
d, steps = sys.argv[1], int(sys.argv[2])
db = glob.glob(d + '/*/*_results.db')[0]
rows = sqlite3.connect(db).execute('select * from top_kernels').fetchall()      # name, calls, total us, avg us, percent
tot = sum(r[2] for r in rows)
out = [('kernel', 'calls', 'avg_us', 'us_per_step', 'percent')]
for r in rows:
    out.append((re.sub(r'\(.*', '', r[0]), r[1], round(r[3], 2), round(r[2] / steps, 1), round(r[4], 2)))
for o in out[:int(sys.argv[4]) if len(sys.argv) > 4 else 26]:
    print('%-88s %6s %9s %10s %6s' % (o[0][-88:], o[1], o[2], o[3], o[4]))
print('total us/step: %.1f' % (tot / steps))
if len(sys.argv) > 3 and sys.argv[3] != '-':
    csv.writer(open(sys.argv[3], 'w')).writerows(out)
