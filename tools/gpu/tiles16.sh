# volume-dependent block tiles of the 16-bit layer kernels: 16-bit parity (small shapes + cfg3 / cfg5 full size), bench
cd $GRAFT_REPO_ROOT
python tests/layer_check.py 16 24 40 f16 bf16 > gpurun_out/t16_layer_check.log 2>&1; echo "layer_check rc=$?"; tail -1 gpurun_out/t16_layer_check.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "cfg3 or cfg5 or 16bit or bf16 or f16 or storage" 2>&1 | tail -3
python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], {k:(v['value'], {s:v['stages_ms'][s] for s in ('conv4','conv5','conv6','conv7','conv9')}) for k,v in d['other_configs'].items()})"
