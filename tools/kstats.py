"""Aggregate a rocprofv3 kernel_stats csv into per-frame numbers: python tools/kstats.py stats.csv n_frames"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
out = []
for r in rows:
    name = r["Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:60]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    out.append((t, int(r["Calls"]), float(r["AverageNs"]) / 1e3, name))
print(f"total kernel time {tot/frames:.1f} us/frame")
for t, c, a, n in sorted(out, reverse=True)[:22]:
    print(f"{t/frames:9.1f} us/frame  {c/frames:7.1f} calls/frame  {a:8.2f} us avg  {n}")
