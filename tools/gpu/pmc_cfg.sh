# the three PMC passes of tools/gpu/session.sh for another BASELINE config: bash tools/gpu/pmc_cfg.sh cfg3 bf16
# -> gpurun_out/traffic_$1 (merge with: python tools/pmc_summary.py gpurun_out/traffic_cfg3 cfg3 bf16 > profiles/rNN_traffic_cfg3.json)
set -e
R=$GRAFT_REPO_ROOT
C=${1:-cfg3}
T=${2:-bf16}
export TMPDIR=/tmp
rm -rf $R/gpurun_out/traffic_$C
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
n=$(echo $pass | cut -d' ' -f1)
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$C/$n -- python3 $R/tools/prof_stage.py all 2 $C $T > /dev/null 2>&1
done
echo PMC_DONE
