# kernel-only durations of the fused conv11+prob kernel: default build and the diagnostic ablation builds
# (make -C scene_3dreconstruction_mvsnet_amd/csrc ablate21 .. ablate24; they act on the first form: MVS_FUSE_PROB_FORM=1)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/fab
cd /tmp
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fab/$1 -- python3 $R/tools/prof_stage.py all 10 > /dev/null 2>&1
  f=$(find $R/gpurun_out/fab/$1 -name '*kernel_stats.csv' | head -1)
  python3 - "$1" "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'conv11_prob' in r['Name']:
        print(sys.argv[1], r['Name'][:36], r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
}
run default
export MVS_FUSE_PROB_FORM=1
for n in 21 22 23 24; do
  export MVS_LIB_PATH=$R/scene_3dreconstruction_mvsnet_amd/csrc/libmvs_hip_ablate$n.so
  run ablate$n
done
