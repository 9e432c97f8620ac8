# GPU session script (run on the box through gpurun): A/B timings, tests, bench, PMC passes
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/gpu/dbg_warp.py > gpurun_out/dbg_warp.txt 2>&1; grep "^case" gpurun_out/dbg_warp.txt
{ for f in 1 2; do for nt in 0 1; do echo "TC=$f NT=$nt"; MVS_WARP_TC=$f MVS_WARP_NT=$nt python tools/time_stage.py warp 50; done; done
  for z in 0 8 16 32; do echo "ZMARCH=$z"; MVS_PROB_ZMARCH=$z python tools/time_stage.py prob 50; done
  python tools/time_stage.py conv0 50; python tools/time_stage.py conv0z 50; } > gpurun_out/r2_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2_ab.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t1.log 2>&1 || { tail -60 gpurun_out/r2_t1.log; exit 1; }
tail -3 gpurun_out/r2_t1.log
python bench.py --steps 60 > gpurun_out/r2_b1.json 2> gpurun_out/r2_b1.err
python bench.py --steps 60 --streams 2 --no-cpu-baseline --no-e2e > gpurun_out/r2_b1s2.json 2>> gpurun_out/r2_b1.err
MVS_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-e2e > gpurun_out/r2_b1g2.json 2> gpurun_out/r2_b1g2.err || true
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq1 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py all 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq2 -- python3 $GRAFT_REPO_ROOT/tools/prof_stage.py all 3 > /dev/null 2>&1
echo PMC_DONE
