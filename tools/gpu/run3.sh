# quick warp-kernel session: correctness of the forms, A/B timing, masked-gather probe, short bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/gpu/dbg_warp.py > gpurun_out/dbg_warp.txt 2>&1; grep "^case" gpurun_out/dbg_warp.txt
{ for f in 2 3 2 3; do echo "FORM=$f"; MVS_WARP_TC=$f python tools/time_stage.py warp 50; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ab3.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "warp or variance or depth_infer or full_size or per_stage" 2>&1 | tail -3
python bench.py --steps 40 --no-cpu-baseline --no-e2e > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err
MVS_WARP_TC=2 python bench.py --steps 40 --no-cpu-baseline --no-e2e > gpurun_out/r2_b3nt0.json 2>> gpurun_out/r2_b3.err
