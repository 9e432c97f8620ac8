cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "conv11 or fused or maps" 2>&1 | tail -2
python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['stages']['conv11_prob']['ms'], {k:(v['value'], v['stages_ms']['conv11_prob']) for k,v in d['other_configs'].items()})"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm/$c -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py all 2 > /dev/null 2>&1; done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("/tmp/pm/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv11_prob" in r["Kernel_Name"]: tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v)/len(v) for k, v in tot.items()})
PY
