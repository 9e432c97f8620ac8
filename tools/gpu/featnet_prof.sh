# FeatureNet (HIP) at the cfg2 image size: HIP-event time and rocprofv3 per-kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/prof_featnet.py 50 2>&1 | grep -v amdgpu.ids
python tools/prof_featnet.py 50 u8 2>&1 | grep -v amdgpu.ids
cd /tmp
rm -rf /tmp/fn
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fn -- python3 $GRAFT_REPO_ROOT/tools/prof_featnet.py 20 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/fn/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), r["Percentage"])
PY
