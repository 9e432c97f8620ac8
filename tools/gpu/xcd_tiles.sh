# XCD-aware tile order in the tile kernels (default lib) against plain blockIdx order (libmvs_hip_ablate99.so = the build
# before the change): 16-bit parity, stage times at cfg3 / cfg5
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
python tests/layer_check.py 16 24 40 f16 bf16 > gpurun_out/xcd_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -1 gpurun_out/xcd_layer_check.log
for l in libmvs_hip.so libmvs_hip_ablate99.so libmvs_hip.so libmvs_hip_ablate99.so; do
MVS_LIB_PATH=$C/$l python bench.py --config cfg3 --streams 1 --steps 8 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/b3.json 2>/dev/null
MVS_LIB_PATH=$C/$l python bench.py --config cfg5 --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/b5.json 2>/dev/null
python -c "
import json
for f in ('/tmp/b3.json','/tmp/b5.json'):
    d=json.load(open(f)); print('$l', d['value'], {k: v['ms'] for k, v in d['stages'].items() if k in ('conv4','conv5','conv6','conv7','conv9')})"
done
