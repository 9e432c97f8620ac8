# PMC counters of the conv0 kernel alone (tools/prof_stage.py conv0, whatever MVS_CONV0_SPLIT selects): three passes
# -> gpurun_out/conv0_pmc/*.csv, per-launch averages printed
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/conv0_pmc
cd /tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
i=$((i+1))
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/conv0_pmc/p$i -- python3 $R/tools/prof_stage.py conv0 3 > $R/gpurun_out/conv0_pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/conv0_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv0" not in k: continue
        tot[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, d in tot.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
