"""Ordered kernel list of the last decode frame from a rocprofv3 --kernel-trace CSV:
   python tools/ktrace.py <kernel_trace.csv> [n_last_dispatches]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 520
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:70]
    grid = f'{r.get("Grid_Size_X", r.get("Grid_Size", "?"))}x{r.get("Grid_Size_Y", "")}x{r.get("Grid_Size_Z", "")}/{r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))}'
    gap = (a - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(a-t0)/1e3:9.2f} {(b-a)/1e3:7.2f} {gap:7.2f}  {grid:22s} {name}")
    prev_end = b
