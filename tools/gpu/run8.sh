set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ for p in 0 1 0 1; do echo "PERSIST=$p"; MVS_PROB_PERSIST=$p python tools/time_stage.py prob 100; done; } 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "layer or depth_infer or cost_volume" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r2_b8.json 2> gpurun_out/r2_b8.err
python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/r2_b8s1.json 2>> gpurun_out/r2_b8.err
