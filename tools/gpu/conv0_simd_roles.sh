# conv0_w43p with the consumer waves on two SIMDs and the producers on the other two (`make ablate75`, correct results)
# against the default (one consumer and one producer per SIMD)
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
MVS_LIB_PATH=$C/libmvs_hip_ablate75.so python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "conv0" 2>&1 | tail -1
for r in 1 2; do
python tools/time_stage.py conv0 300 2>&1 | grep -v amdgpu.ids
MVS_LIB_PATH=$C/libmvs_hip_ablate75.so python tools/time_stage.py conv0 300 2>&1 | grep -v amdgpu.ids
done
