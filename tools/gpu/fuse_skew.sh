# fused conv11+prob kernel: kernel-only durations per variant (environment switches of conv11_prob.hip)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/fsk
cd /tmp
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fsk/$1 -- python3 $R/tools/prof_stage.py all 10 > /dev/null 2>&1
  f=$(find $R/gpurun_out/fsk/$1 -name '*kernel_stats.csv' | head -1)
  python3 - "$1" "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'conv11_prob' in r['Name']:
        print(sys.argv[1], r['Name'][:40], r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
}
run form2
export MVS_FUSE_PROB_ZC=8; run form2_zc8; unset MVS_FUSE_PROB_ZC
export MVS_FUSE_PROB_FORM=1; run form1; unset MVS_FUSE_PROB_FORM
