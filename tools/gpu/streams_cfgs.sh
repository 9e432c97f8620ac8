# sustained rate vs number of streams (cfg2), and the other BASELINE configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 1 2 3 4; do
  python bench.py --streams $s --steps 600 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-e2e --staged-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s ->', d['value'], 'maps/s')"
done
python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5.json 2>/dev/null
python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/bench_cfg3.json 2>/dev/null
python bench.py --config cfg1 --no-cpu-baseline --no-e2e > gpurun_out/bench_cfg1.json 2>/dev/null
for c in cfg5 cfg3 cfg1; do python3 -c "
import json
d=json.loads(open('gpurun_out/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['first_pass']['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if v['ms']>0.05})"; done
