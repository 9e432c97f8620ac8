# full session: tests, bench (default + variants), dataset feed, rocprof stats + PMC traffic
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t6.log 2>&1 || { tail -60 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
python bench.py > gpurun_out/r2_b6.json 2> gpurun_out/r2_b6.err
python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/r2_b6s1.json 2>> gpurun_out/r2_b6.err
python bench.py --streams 3 --no-cpu-baseline --no-e2e > gpurun_out/r2_b6s3.json 2>> gpurun_out/r2_b6.err
for c in cfg5 cfg3; do
python bench.py --config $c --no-cpu-baseline --no-e2e --steps 20 > gpurun_out/r2_b6_$c.json 2>> gpurun_out/r2_b6.err
MVS_WARP_TC16=1 python bench.py --config $c --no-cpu-baseline --no-e2e --steps 20 > gpurun_out/r2_b6_${c}_tc16.json 2>> gpurun_out/r2_b6.err
done
python tools/time_dataset_driver.py 98 > gpurun_out/r2_dataset.txt 2>&1 || tail -20 gpurun_out/r2_dataset.txt
grep -v amdgpu.ids gpurun_out/r2_dataset.txt | tail -8
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e > /dev/null 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
n=$(echo $pass | cut -d' ' -f1)
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_traffic/$n -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py all 3 > /dev/null 2>&1
done
echo SESSION_DONE
