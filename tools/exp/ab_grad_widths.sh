cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "" "--grad-f16 1" "--vit-path stream16=1" "--grad-f16 1 --vit-path stream16=1"; do
  python bench.py --steps 40 --no-cpu-baseline --no-legs --no-roofline $v 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.1f steps/s  %s' % ('$v' or 'default', j['value'], ['%.1f' % x for x in j['repeats']['steps_per_s']]))"
done; done
