#!/usr/bin/env python3
"""Merge rocprofv3 PMC passes into profiles/rNN_traffic.json (per-kernel averages per launch).

Collect (each pass separately, counters never mixed with sys/hip traces):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 tools/prof_stage.py all 3
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python3 tools/prof_stage.py all 3
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \\
              SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d OUT/sq -- python3 ...
Then:  python tools/pmc_summary.py OUT [cfg2|cfg3|cfg5] [f32|f16|bf16] > profiles/rNN_traffic[_cfgN].json
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per count as rocprofv3 reports them; the gfx950
correction (FETCH_SIZE under-reports wide 16 B/lane streaming reads by 2x, MI355X_MICROARCH.md HBM
section) is applied in `hbm_bytes_fetch_x2`.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
storage = sys.argv[3] if len(sys.argv) > 3 else {"cfg3": "bf16", "cfg5": "f16"}.get(cfg_name, "f32")
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name")
            if not name or "mvs::" not in name:
                continue
            short = name.rsplit("(", 1)[0].replace("void ", "").replace("(anonymous namespace)::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, each with "
                 f"--kernel-trace only) over tools/prof_stage.py all 3 {cfg_name} {storage}, MI355X; merged by "
                 f"tools/pmc_summary.py OUT {cfg_name} {storage}",
       "note": "FETCH_SIZE under-reports wide coalesced 16 B/lane streaming reads by 2x on gfx950 "
               "(MI355X_MICROARCH.md, HBM section); hbm_bytes_fetch_x2 applies that correction, valid for "
               "the staging loads of the conv kernels, not for the tap gathers of the warp kernel.  "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles.",
       "kernels": {}}
for k, ctrs in sorted(acc.items()):
    ent = {}
    avg = {c: sum(v) / len(v) for c, v in ctrs.items()}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        ent["FETCH_SIZE_KB_avg"] = round(avg["FETCH_SIZE"], 1)
        ent["WRITE_SIZE_KB_avg"] = round(avg["WRITE_SIZE"], 1)
        ent["hbm_bytes_uncorrected"] = int((avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        ent["hbm_bytes_fetch_x2"] = int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
    sq = {c: int(v) for c, v in avg.items() if c.startswith("SQ_")}
    if sq:
        ent["sq"] = sq
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
            # summed over the chip's 1024 SIMDs (256 CUs x 4): busy cycles of one SIMD's matrix pipe
            ent["mfma_busy_cycles_per_simd"] = int(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024)
    out["kernels"][k] = ent
print(json.dumps(out, indent=1))
