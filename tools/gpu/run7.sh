set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out


python -m pytest tests -m gpu -x -q > gpurun_out/r2_t7.log 2>&1 || { tail -60 gpurun_out/r2_t7.log; exit 1; }
tail -3 gpurun_out/r2_t7.log
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r2_b7.json 2> gpurun_out/r2_b7.err
for c in cfg5 cfg3; do python bench.py --config $c --no-cpu-baseline --no-e2e --steps 20 > gpurun_out/r2_b7_$c.json 2>> gpurun_out/r2_b7.err; done
python tools/time_dataset_driver.py 98 > gpurun_out/r2_dataset.txt 2>&1 || tail -20 gpurun_out/r2_dataset.txt
grep -v amdgpu.ids gpurun_out/r2_dataset.txt | tail -7
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_traffic7/$pass -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py conv0 3 > /dev/null 2>&1
done
echo SESSION_DONE
