# per-kernel averages of the drop-in MVSNet.forward from images (FeatureNet in HIP + path) under rocprofv3
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/e2e
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e2e/s -- python3 $R/tools/gpu/e2e_loop.py 200 > $R/gpurun_out/e2e/out.txt 2>&1
f=$(find $R/gpurun_out/e2e/s -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/e2e/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = max(int(r['Calls']) for r in rows if 'conv0_w43' in r['Name'])
tot = 0
for r in rows:
    per_map = float(r['TotalDurationNs']) / n / 1e3
    if per_map > 3:
        print(f"{r['Name'][:80]:80s} calls/map {int(r['Calls'])/n:5.1f}  us/map {per_map:8.1f}")
    tot += per_map
print('sum us/map', round(tot, 1), 'maps', n)
PY
tail -2 $R/gpurun_out/e2e/out.txt
