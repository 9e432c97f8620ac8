# split-operand transposed layers conv7 / conv9 (MVS_SPLIT_DECONV bit 0 / bit 1) against the fp32-MFMA kernels: parity
# (small shapes + cfg2 per layer), bench stage times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MVS_SPLIT_DECONV=3 python tests/layer_check.py 16 24 40 > gpurun_out/sd_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -2 gpurun_out/sd_layer_check.log
MVS_SPLIT_DECONV=3 python tests/layer_check.py 24 40 56 > gpurun_out/sd_layer_check2.log 2>&1; echo "layer_check2 rc=$?"; tail -1 gpurun_out/sd_layer_check2.log
MVS_SPLIT_DECONV=3 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "every_layer_matches_oracle or cost_volume_and_maps or reference_fixture" -s > gpurun_out/sd_fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/sd_fullsize.log | head -20
for v in 3 0 3 0; do
MVS_SPLIT_DECONV=$v python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > gpurun_out/sd_bench$v.json 2> gpurun_out/sd_bench$v.err
python - <<PY
import json
d = json.load(open("gpurun_out/sd_bench$v.json"))
print("MVS_SPLIT_DECONV=$v", d["value"], {k: v["ms"] for k, v in d["stages"].items() if k in ("conv7", "conv9")})
PY
done
