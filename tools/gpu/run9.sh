set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dataset or save_depth" 2>&1 | tail -3
MVS_DRIVER_TRACE=1 python tools/time_dataset_driver.py 196 > gpurun_out/r2_dataset.txt 2>&1 || tail -20 gpurun_out/r2_dataset.txt
grep -v amdgpu.ids gpurun_out/r2_dataset.txt | tail -9
cd /tmp
echo DONE
