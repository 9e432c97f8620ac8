# warp + variance with four waves per SIMD requested (default) against `make ablate61` (no occupancy request: 130 VGPRs for
# N = 5 in fp32): bit-identity of the two-stream probes, the warp tests, stage times at the three bench configs
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
echo "pair test: clean reps of 8:"; python tools/probes/pair_two_streams.py 48 32 40 8 2>&1 | grep -v amdgpu.ids | grep -c "'warp': 0, 'tail': 0"
echo "path test: mismatching iterations:"; MVS_CONV0_SPLIT=0 python tools/probes/path_two_streams.py 48 32 40 4 2>&1 | grep -v amdgpu.ids | awk '{s+=$6} END {print s, "of 320"}'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "tap_cache or two_host or warp or variance or in_flight" 2>&1 | tail -2
for l in libmvs_hip.so libmvs_hip_ablate61.so libmvs_hip.so libmvs_hip_ablate61.so; do
MVS_LIB_PATH=$C/$l python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); print('$l', d['value'], d['stages']['warp_variance']['ms'], {k:(v['value'], v['stages_ms']['warp_variance']) for k,v in d['other_configs'].items()})"
done
