# PMC counters of the fused conv11+prob kernel (two passes), printed per kernel as averages over the launches
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/fpmc
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
i=$((i+1))
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/fpmc/p$i -- python3 $R/tools/prof_stage.py all 3 > /dev/null 2>&1 || echo "pass $i failed"
done
python3 - $R/gpurun_out/fpmc <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv11_prob' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k][0] / acc[k][1]:16.0f}  (n={acc[k][1]})")
PY
