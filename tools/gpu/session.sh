# One GPU session (run on the box through gpurun: `gpurun --timeout 1100 -- 'bash tools/gpu/session.sh'`):
# the GPU test suite, the default bench line + the single-stream line, the rocprofv3 kernel stats of the
# single-stream command and the three PMC passes that tools/pmc_summary.py merges into profiles/rNN_traffic.json.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/traffic gpurun_out/stats_1stream
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench.err
python bench.py --streams 1 --no-cpu-baseline --no-e2e > gpurun_out/bench_1stream.json 2>> gpurun_out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats_1stream -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/bench_1stream_under_rocprof.json 2>/dev/null
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
n=$(echo $pass | cut -d' ' -f1)
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/$n -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py all 3 > /dev/null 2>&1
done
echo SESSION_DONE1
# N > 1 plumbing on the one-GPU box: 4 ranks on cuda:0 over gloo (the box allows at most 6 processes on its GPU; the
# launcher's agent process opens it too, so 6 ranks trip the guard -- measured in round 4)
cd $GRAFT_REPO_ROOT
MVS_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --steps 4 --no-cpu-baseline --no-e2e > gpurun_out/rehearsal4.json 2> gpurun_out/rehearsal4.err || echo "rehearsal rc=$?"
echo SESSION_DONE2
