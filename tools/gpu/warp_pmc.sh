# PMC counters of the warp + variance kernel alone (tools/prof_stage.py warp): two passes -> gpurun_out/warp_pmc/*.csv
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/warp_pmc
cd /tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
i=$((i+1))
rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/warp_pmc/p$i -- python3 $R/tools/prof_stage.py warp 3 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/warp_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "warp_variance" not in k: continue
        tot[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, d in tot.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
