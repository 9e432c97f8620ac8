# timing-only ablation builds of the split-operand tail (wrong results by design): 81 no stencil, 82 no MFMAs, 83 no split
# arithmetic, 84 no scatter / skip add, 85 / 86 one B / A fragment read per chunk and piece
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
for r in 1 2; do
python tools/time_stage.py tail 300
for a in $(ls $C | sed -n 's/libmvs_hip_ablate\([0-9]*\).so/\1/p'); do MVS_LIB_PATH=$C/libmvs_hip_ablate$a.so python tools/time_stage.py tail 300; done
done 2>&1 | grep -v amdgpu.ids
