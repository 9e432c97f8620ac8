set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t12.log 2>&1 || { tail -60 gpurun_out/r2_t12.log; exit 1; }
tail -3 gpurun_out/r2_t12.log
python bench.py > gpurun_out/r2_b12.json 2> gpurun_out/r2_b12.err
python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/r2_b12s1.json 2>> gpurun_out/r2_b12.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_stats1 -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/r2_b12s1_rocprof.json 2>/dev/null
echo DONE
