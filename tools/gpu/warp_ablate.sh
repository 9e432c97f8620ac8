# kernel-only durations of the warp+variance kernel for the default build, MVS_WARP_PAIR=0 and the
# diagnostic ablation builds (make -C scene_3dreconstruction_mvsnet_amd/csrc ablate11 .. ablate14)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/wab
cd /tmp
run() {  # name, env assignments are exported by the caller
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wab/$1 -- python3 $R/tools/prof_stage.py warp 30 > /dev/null 2>&1
  f=$(find $R/gpurun_out/wab/$1 -name '*kernel_stats.csv' | head -1)
  echo "$1: $(grep warp_variance_tc2 $f | cut -d, -f1-5 | tr '\n' ' ' | cut -c1-30,120-)"
}
run default
export MVS_WARP_PAIR=0; run pair0; unset MVS_WARP_PAIR
for n in 11 12 13 14; do
  export MVS_LIB_PATH=$R/scene_3dreconstruction_mvsnet_amd/csrc/libmvs_hip_ablate$n.so
  run ablate$n
  export MVS_WARP_PAIR=0; run ablate${n}_pair0; unset MVS_WARP_PAIR
done
