# quick warp-kernel session: correctness of the forms, A/B timing, masked-gather probe, short bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/gpu/dbg_warp.py > gpurun_out/dbg_warp.txt 2>&1; grep "^case" gpurun_out/dbg_warp.txt
{ echo "form1"; MVS_WARP_TC=1 python tools/time_stage.py warp 50
  for c in 8 4; do for nt in 0 1; do for sl in 24 48; do echo "CPT=$c NT=$nt SLAB=$sl"; MVS_WARP_CPT=$c MVS_WARP_NT=$nt MVS_WARP_TC_SLAB=$sl python tools/time_stage.py warp 50; done; done; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ab3.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "warp or variance or depth_infer or full_size or per_stage" 2>&1 | tail -3
python bench.py --steps 40 --no-cpu-baseline --no-e2e > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err
MVS_WARP_CPT=8 python bench.py --steps 40 --no-cpu-baseline --no-e2e > gpurun_out/r2_b3nt0.json 2>> gpurun_out/r2_b3.err
