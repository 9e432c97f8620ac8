# PMC passes for the warp kernel (default form) + CPT/slab sweep
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ for sl in 24 48 96 192; do echo "SLAB=$sl"; MVS_WARP_TC_SLAB=$sl python tools/time_stage.py warp 50; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ab4.txt
cd /tmp
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_c4_sq1 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_c4_sq2 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_c4_tcp -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_c4_fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_c4_write -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py warp 3 > /dev/null 2>&1 || true
echo PMC_DONE
