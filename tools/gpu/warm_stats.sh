# per-kernel averages of a long (warm-clock) single-stream run under rocprofv3 --kernel-trace --stats
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/warm
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/warm/s1 -- python3 $R/bench.py --streams 1 --steps 600 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-e2e --staged-steps 0 > $R/gpurun_out/warm/bench.json 2>/dev/null
f=$(find $R/gpurun_out/warm/s1 -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 600:
        print(f"{r['Name'][:72]:72s} {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
        tot += float(r['AverageNs']) / 1e3
print('sum', round(tot, 1))
PY
tail -c 300 $R/gpurun_out/warm/bench.json | head -c 10; python3 -c "
import json; d=json.loads(open('$R/gpurun_out/warm/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
