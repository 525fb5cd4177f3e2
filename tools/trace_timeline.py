#!/usr/bin/env python3
"""Timeline of the last graph replay in a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous end, grid, kernel."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", r["Kernel_Name"])[:60]
    print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:7.1f} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{r['Workgroup_Size_X']:>4s} {name}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us")
