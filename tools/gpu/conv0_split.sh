# conv0 with split bf16 operands (default) against the fp32-MFMA kernel (MVS_CONV0_SPLIT=0) and the split kernel's first form
# (=2): parity with UNCHANGED bounds, kernel time, bench line.  gpurun --timeout 900 -- 'bash tools/gpu/conv0_split.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/layer_check.py 16 24 40 > gpurun_out/split_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -1 gpurun_out/split_layer_check.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "every_layer_matches_oracle or heavy_tailed or cost_volume_and_maps or reference_fixture" -s > gpurun_out/split_fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "heavy-tailed|passed|failed|Error|assert" gpurun_out/split_fullsize.log | head -20
for i in 1 2; do
MVS_CONV0_SPLIT=2 python tools/time_stage.py conv0 200
MVS_CONV0_SPLIT=0 python tools/time_stage.py conv0 200
MVS_CONV0_SPLIT=1 python tools/time_stage.py conv0 200
done
python bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > gpurun_out/split_bench.json 2> gpurun_out/split_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/split_bench.json"))
print("bench", d["value"], d["first_pass"]["value"], d["dtype"], {k: v["ms"] for k, v in d["stages"].items()})
PY
