# LDS bank-conflict share of every kernel of the depth path (cfg2 fp32, cfg3 bf16): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
# and the LDS-busy share of the kernel's cycles (SQ_LDS_IDX_ACTIVE / 256 CUs against SQ_BUSY_CYCLES / 32 shader engines)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/lds_scan
cd /tmp
for cfg in cfg2 cfg3; do
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/lds_scan/$cfg -- python3 $R/tools/prof_stage.py all 2 $cfg > $R/gpurun_out/lds_scan_$cfg.log 2>&1 || echo "$cfg failed"
done
python3 - <<PY
import csv, glob, collections
for cfg in ("cfg2", "cfg3"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/lds_scan/%s/*/*counter_collection.csv" % cfg):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print(cfg)
    for k, d in tot.items():
        a = {c: v / n[(k, c)] for c, v in d.items()}
        idx, bc, busy = a.get("SQ_LDS_IDX_ACTIVE", 0), a.get("SQ_LDS_BANK_CONFLICT", 0), a.get("SQ_BUSY_CYCLES", 1)
        if idx > 0:
            print("  %-70s conflict %.2f  lds_busy %.2f  wait_inst_lds %.2f" % (k, bc / idx, (idx / 256) / (busy / 32), a.get("SQ_WAIT_INST_LDS", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)))
PY
