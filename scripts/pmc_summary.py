#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<TAG>_*/…/*_counter_collection.csv): per-launch
mean of every counter for kernels whose name contains PATTERN, plus the derived numbers bench.py
reports in `roofline_valu` / `roofline.traffic`.

    scripts/pmc_summary.py TAG [PATTERN] [WORKLOAD PRECISION KERNEL_USED]
"""
import csv
import glob
import json
import sys
from collections import defaultdict

tag = sys.argv[1]
pattern = sys.argv[2] if len(sys.argv) > 2 else "rtow_trace"
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv")):
    acc = defaultdict(lambda: defaultdict(float))
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pattern not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            meta = {k: row[k] for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Grid_Size",
                                        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size") if k in row}
    for name, per in acc.items():
        vals = list(per.values())
        out[name] = sum(vals) / len(vals)
    out.setdefault("_meta", {}).update(meta)
if len(sys.argv) > 5:
    out["workload"], out["precision"], out["kernel_used"] = sys.argv[3], sys.argv[4], int(sys.argv[5])
g = out.get
d = {}
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    # SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
    d["valu_issue_utilisation"] = round(g("SQ_ACTIVE_INST_VALU") * 4 / (g("GRBM_GUI_ACTIVE") / 8 * 1024), 4)
if g("SQ_INSTS_VALU"):
    d["valu_insts_per_launch"] = int(g("SQ_INSTS_VALU"))
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
    d["lane_activity"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_INSTS_VALU") * 64), 4)
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    # KB per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)
    d["hbm_bytes_per_launch"] = int(g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024)
    d["hbm_bytes_per_launch_raw"] = int(g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    d["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
    d["wave_wait_any_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
    d["wave_wait_inst_frac"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
out["derived"] = d
# which kernel these counters belong to: bench.py reports them only for the same sources
import hashlib, pathlib
_h = hashlib.sha1()
_d = pathlib.Path(__file__).resolve().parent.parent / "raytracing-one-weekend_amd" / "csrc"
for _f in sorted(list(_d.glob("rtow_trace_*.h")) + [_d / "rtow_device.h"]):
    _h.update(_f.read_bytes())
out["kernel_source_sha"] = _h.hexdigest()[:16]
print(json.dumps(out, indent=1))
