# conv0z16 with the kz = 1 fragments of the first C chunks cached in registers (record of the round-4 A/B: the default was
# C = 2 at the time, libmvs_hip_ablate90 / 93 / 94 were builds with -DC0Z_CACHE_C=0 / 3 / 4 of conv3d_mfma16.hip; default now 4):
# 16-bit parity (identical results by construction: same order), stage times at cfg3 / cfg5
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
for l in libmvs_hip.so libmvs_hip_ablate94.so; do
MVS_LIB_PATH=$C/$l python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "cfg3 or cfg5 or 16bit or conv0" 2>&1 | tail -1
done
for l in libmvs_hip_ablate90.so libmvs_hip.so libmvs_hip_ablate93.so libmvs_hip_ablate94.so libmvs_hip_ablate90.so libmvs_hip.so libmvs_hip_ablate93.so libmvs_hip_ablate94.so; do
MVS_LIB_PATH=$C/$l python bench.py --config cfg3 --streams 1 --steps 8 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/b3.json 2>/dev/null
MVS_LIB_PATH=$C/$l python bench.py --config cfg5 --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/b5.json 2>/dev/null
python -c "
import json; a=json.load(open('/tmp/b3.json')); b=json.load(open('/tmp/b5.json')); print('$l cfg3', a['value'], a['stages']['conv0']['ms'], 'cfg5', b['value'], b['stages']['conv0']['ms'])"
done
