# conv tile kernels (convgs fp32 split, convg16 16-bit) with the LDS-staged epilogue (default lib) against the scalar epilogue
# (libmvs_hip_ablate99.so = the build before): parity, stage times
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
python tests/layer_check.py 16 24 40 f32 f16 bf16 > gpurun_out/ce_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -1 gpurun_out/ce_layer_check.log
python tests/layer_check.py 24 40 56 f32 bf16 > gpurun_out/ce_layer_check2.log 2>&1; echo "layer_check2 rc=$?"; tail -1 gpurun_out/ce_layer_check2.log
MVS_CONVZ16=0 python tests/layer_check.py 16 24 40 f16 > gpurun_out/ce_layer_check3.log 2>&1; echo "layer_check3 rc=$?"; tail -1 gpurun_out/ce_layer_check3.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "every_layer_matches_oracle or cfg3 or cfg5 or heavy" 2>&1 | tail -1
for l in libmvs_hip.so libmvs_hip_ablate99.so libmvs_hip.so libmvs_hip_ablate99.so; do
MVS_LIB_PATH=$C/$l python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); L=('conv2','conv3','conv4','conv5','conv6'); print('$l', d['value'], {k: v['ms'] for k, v in d['stages'].items() if k in L}, {k:(v['value'], [v['stages_ms'][s] for s in L]) for k,v in d['other_configs'].items()})"
done
