# PMC counters of the fused tail alone (tools/prof_stage.py all -> the conv11_prob kernel rows; MVS_LIB_PATH / MVS_TAIL_SPLIT
# select the variant): four passes -> gpurun_out/tail_pmc/*.csv, per-launch averages printed
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/tail_pmc
cd /tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
i=$((i+1))
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/tail_pmc/p$i -- python3 $R/tools/prof_stage.py all 3 > $R/gpurun_out/tail_pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/tail_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv11_prob" not in k: continue
        tot[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, d in tot.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
