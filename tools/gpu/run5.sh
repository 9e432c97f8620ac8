set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
{ echo "full"; python tools/time_stage.py warp 50
  for v in 11 12 13; do echo "ablate $v"; MVS_LIB_PATH=$C/libmvs_hip_ablate$v.so python tools/time_stage.py warp 50; done
  echo "full"; python tools/time_stage.py warp 50; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ab5.txt
