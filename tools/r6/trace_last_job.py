#!/usr/bin/env python3
"""Print the kernels of the LAST multiexp in a rocprofv3 --kernel-trace csv (start offset, duration, grid, name), i.e.
everything after the last msm_digits_kernel launch.   usage: trace_last_job.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "msm_digits_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bh::", "")[:58]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  grid %8s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), name))
