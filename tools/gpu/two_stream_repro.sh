# bit-identity of the path's stages when two host threads drive two HIP streams (tools/probes/*_two_streams.py), the tests
# that watch it, and the warp kernel's time at the three bench configs
cd $GRAFT_REPO_ROOT
echo "pair test: clean reps of 12:"; python tools/probes/pair_two_streams.py 48 32 40 12 2>&1 | grep -v amdgpu.ids | grep -c "'warp': 0, 'tail': 0"
echo "path test: mismatching iterations:"; MVS_CONV0_SPLIT=0 python tools/probes/path_two_streams.py 48 32 40 5 2>&1 | grep -v amdgpu.ids | awk '{s+=$6} END {print s, "of 400"}'
python tools/probes/path_two_streams.py 48 32 40 5 2>&1 | grep -v amdgpu.ids | awk '{s+=$6} END {print s, "of 400"}'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "tap_cache or two_host or warp or variance or in_flight" 2>&1 | tail -2
python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['stages']['warp_variance']['ms'], {k:(v['value'], v['stages_ms']['warp_variance']) for k,v in d['other_configs'].items()})"
