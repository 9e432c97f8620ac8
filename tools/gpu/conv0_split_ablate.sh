# timing-only ablation builds of the persistent split conv0 (wrong results by design): 71 no MFMA phase, 72 no activation
# loads after the prologue, 73 no LDS writes of the split pieces, 74 no split arithmetic
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
for r in 1 2; do
python tools/time_stage.py conv0 300
for a in 71 72 73 74; do MVS_LIB_PATH=$C/libmvs_hip_ablate$a.so python tools/time_stage.py conv0 300; done
done
