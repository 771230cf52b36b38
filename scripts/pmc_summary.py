#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<TAG>_*/…/*_counter_collection.csv):
per-launch mean of every counter for kernels whose name contains PATTERN."""
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
            meta = {k: row[k] for k in ("VGPR_Count", "SGPR_Count", "Grid_Size", "Workgroup_Size",
                                        "LDS_Block_Size", "Scratch_Size") if k in row}
    for name, per in acc.items():
        vals = list(per.values())
        out[name] = sum(vals) / len(vals)
    out.setdefault("_meta", {}).update(meta)
print(json.dumps(out, indent=1))
