# the three PMC passes of tools/gpu/session.sh alone (-> gpurun_out/traffic, merged by tools/pmc_summary.py)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/traffic
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
n=$(echo $pass | cut -d' ' -f1)
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$n -- python3 $R/tools/prof_stage.py all 3 > /dev/null 2>&1
done
echo PMC_DONE
