# how the bench value depends on the number of timed / warm-up steps (clock ramp, pipeline fill and drain)
cd $GRAFT_REPO_ROOT
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 300" "--steps 100 --warmup 5" "--steps 300 --warmup 5" "--steps 1000 --warmup 5" "--steps 3000 --warmup 5" "--steps 20 --warmup 5 --streams 1" "--steps 1000 --warmup 5 --streams 1"; do
  python bench.py $args --no-cpu-baseline --no-e2e --staged-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', '->', d['value'], 'maps/s', d['ms_per_step'], 'ms')"
done
