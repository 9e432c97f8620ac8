# A/B: whole library built with -fno-slp-vectorize (no compiler-made packed fp32 VALU beside the MFMAs).  Build first:
#   make -C scene_3dreconstruction_mvsnet_amd/csrc ablate0 HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -fno-slp-vectorize"
#   mv scene_3dreconstruction_mvsnet_amd/csrc/libmvs_hip_ablate0.so scene_3dreconstruction_mvsnet_amd/csrc/libmvs_hip_noslp.so
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/noslp
cd /tmp
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/noslp/$1 -- python3 $R/tools/prof_stage.py all 10 > /dev/null 2>&1
  f=$(find $R/gpurun_out/noslp/$1 -name '*kernel_stats.csv' | head -1)
  python3 - "$1" "$f" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[2])):
    if float(r['AverageNs']) > 8000 and int(r['Calls']) >= 10:
        print(f"  {sys.argv[1]:8s} {r['Name'][:70]:70s} {float(r['AverageNs'])/1e3:8.1f} us")
        tot += float(r['AverageNs']) / 1e3
print(sys.argv[1], 'sum', round(tot, 1))
PY
}
run default
export MVS_LIB_PATH=$R/scene_3dreconstruction_mvsnet_amd/csrc/libmvs_hip_noslp.so
run noslp
