# split-operand tile kernels for conv2..conv6 (default) against MVS_SPLIT_LAYERS=0: parity (small shapes + cfg2 per layer +
# heavy-tailed), bench stage times of both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/layer_check.py 16 24 40 > gpurun_out/sl_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -2 gpurun_out/sl_layer_check.log
python tests/layer_check.py 24 40 56 > gpurun_out/sl_layer_check2.log 2>&1; echo "layer_check2 rc=$?"; tail -1 gpurun_out/sl_layer_check2.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "every_layer_matches_oracle or heavy_tailed or cost_volume_and_maps or reference_fixture" -s > gpurun_out/sl_fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "heavy-tailed|passed|failed|Error|assert" gpurun_out/sl_fullsize.log | head -20
for v in 1 0; do
MVS_SPLIT_LAYERS=$v python bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > gpurun_out/sl_bench$v.json 2> gpurun_out/sl_bench$v.err
python - <<PY
import json
d = json.load(open("gpurun_out/sl_bench$v.json"))
print("MVS_SPLIT_LAYERS=$v", d["value"], d["first_pass"]["value"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done
