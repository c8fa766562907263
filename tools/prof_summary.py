import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    k = (r["T"], r["N"], r["K"], r["pro"], r["epi"], r["dual"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["us"]); a[2] += float(r["bytes"])
tot = sum(a[1] for a in agg.values())
print(f"total {tot/frames:.1f} us/frame over {len(rows)//frames} launches/frame")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"T={k[0]:>5} N={k[1]:>6} K={k[2]:>6} pro={k[3]} epi={k[4]} dual={k[5]}  n/frame={a[0]/frames:6.1f}  us/launch={a[1]/a[0]:8.2f}  us/frame={a[1]/frames:8.1f}  GB/s={a[2]/a[1]/1e3:8.1f}")
