# XCD-aware tile order in FeatureNet's conv kernels (default lib) against linear order (libmvs_hip_ablate99.so = the build before)
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
python -m pytest tests/test_gpu_featnet.py -m gpu -x -q 2>&1 | tail -1
for l in libmvs_hip.so libmvs_hip_ablate99.so libmvs_hip.so libmvs_hip_ablate99.so; do
echo -n "$l: "; MVS_LIB_PATH=$C/$l python tools/prof_featnet.py 100 2>&1 | grep -v amdgpu.ids
done
