# warp + variance: depths per thread (MVS_WARP_SLAB) and block order, per config
cd $GRAFT_REPO_ROOT
for cfg in cfg3 cfg5 cfg2; do
for slab in 24 20 28 40 12 52 88; do
  MVS_WARP_SLAB=$slab python bench.py --config $cfg --steps 10 --streams 1 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/w.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/w.json')); print('$cfg', 'slab', $slab, 'warp', d['stages']['warp_variance']['ms'])"
done
done
