# warp+variance A/B on the box: parity tests for the warp stage, then HIP-event timings per variant
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "warp or variance or variant or stage or paths" > gpurun_out/warp_tests.log 2>&1 || { tail -40 gpurun_out/warp_tests.log; exit 1; }
tail -2 gpurun_out/warp_tests.log
for i in 1 2; do
MVS_WARP_PAIR=0 python tools/time_stage.py warp 50
MVS_WARP_PAIR=1 python tools/time_stage.py warp 50
done
