# XCD-aware tile order in the split-operand tile kernels (default lib) against plain blockIdx order (libmvs_hip_ablate99.so =
# the build before the change): parity of the layers, stage times
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
python tests/layer_check.py 16 24 40 > gpurun_out/xcd_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -1 gpurun_out/xcd_layer_check.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "every_layer_matches_oracle" 2>&1 | tail -1
for l in libmvs_hip.so libmvs_hip_ablate99.so libmvs_hip.so libmvs_hip_ablate99.so; do
MVS_LIB_PATH=$C/$l python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); print('$l', d['value'], {k: v['ms'] for k, v in d['stages'].items() if k in ('conv2','conv3','conv4','conv9')})"
done
