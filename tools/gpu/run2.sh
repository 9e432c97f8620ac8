# GPU session: tests, bench, PMC passes for both warp forms
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{ for w in 2 4 2 4; do echo "WINO=$w"; MVS_CONV0_WINO=$w python tools/time_stage.py conv0 50; done; } 2>&1 | grep -v amdgpu.ids
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t1.log 2>&1 || { tail -60 gpurun_out/r2_t1.log; exit 1; }
tail -3 gpurun_out/r2_t1.log
python bench.py --steps 60 > gpurun_out/r2_b1.json 2> gpurun_out/r2_b1.err
python bench.py --steps 60 --streams 2 --no-cpu-baseline --no-e2e > gpurun_out/r2_b1s2.json 2>> gpurun_out/r2_b1.err
MVS_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-e2e > gpurun_out/r2_b1g2.json 2> gpurun_out/r2_b1g2.err || true
cd /tmp
for form in 1 2; do
export MVS_WARP_TC=$form
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_f${form}_sq1 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_f${form}_sq2 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_f${form}_tcp -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_f${form}_fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1 || true
done
echo PMC_DONE
