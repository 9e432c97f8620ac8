# timing-only A/B builds of the persistent split conv0 (make ablate56..58: same results, different scheduling)
cd $GRAFT_REPO_ROOT
export MVS_CONV0_SPLIT=1
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
for r in 1 2; do
python tools/time_stage.py conv0 300
for a in 56 57 58; do MVS_LIB_PATH=$C/libmvs_hip_ablate$a.so python tools/time_stage.py conv0 300; done
done
