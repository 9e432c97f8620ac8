set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "layer or full_size_only or depth_infer or costreg or 16bit" 2>&1 | tail -3
python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/r2_b13s1.json 2> gpurun_out/r2_b13.err
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r2_b13.json 2>> gpurun_out/r2_b13.err
MVS_CONV_WINO=0 python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/r2_b13s1_nowino.json 2>> gpurun_out/r2_b13.err
