set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{ python tools/time_stage.py conv1 100; python tools/time_stage.py conv1 100; } 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "layer or full_size_only or depth_infer" 2>&1 | tail -3
MVS_DRIVER_TRACE=1 python tools/time_dataset_driver.py 196 > gpurun_out/r2_dataset.txt 2>&1 || tail -20 gpurun_out/r2_dataset.txt
grep -v "amdgpu.ids\|driver trace" gpurun_out/r2_dataset.txt | tail -12
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r2_b10.json 2> gpurun_out/r2_b10.err
echo DONE
