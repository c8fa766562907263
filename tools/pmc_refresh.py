#!/usr/bin/env python
"""profiles/pmc_traffic.json from a FETCH_SIZE pass of the CURRENT binary.

    python tools/pmc_refresh.py <model-key> <X_pmc_by_kernel.csv> [bench.json of the same build]

<X_pmc_by_kernel.csv> is what `tools/rocprof_summary.py <db> X --pmc` writes for a
`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py ...` run (counters in their own pass, no other trace domain).
FETCH_SIZE is reported in KB and counts 64 B per 128-B request of a wide coalesced stream on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): bytes = value * 1024 * 2.
Per kernel family (vv_gemv_kernel, vv_gemv16p_kernel, vv_attn_fused_kernel, vv_attn_merge2_kernel): dispatches and mean HBM bytes
per launch.  The entry records the library's build id; bench.py carries `roofline.traffic` only when that id equals the id of the
library it runs (a stale file yields null + the reason, never an old number).  Refuses to write when the .so beside the sources
is stale (the pass would not describe HEAD)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vibevoice_amd import build as vbuild  # noqa: E402

FAMILIES = ("vv_gemv_kernel", "vv_gemv16p_kernel", "vv_attn_fused_kernel", "vv_attn_merge2_kernel", "vv_gemm4_kernel",
            "vv_gemm3_kernel", "vv_attn_prefill4_kernel")


def main():
    key, src = sys.argv[1], sys.argv[2]
    bench = sys.argv[3] if len(sys.argv) > 3 else None
    if vbuild.binary_id() != vbuild.source_id():
        raise SystemExit(f"libvvhip.so (build {vbuild.binary_id()}) is not built from the sources beside it ({vbuild.source_id()}): rebuild, re-run the pass")
    agg = {}
    for r in csv.DictReader(open(src)):
        if r["Counter"] != "FETCH_SIZE":
            continue
        name = re.sub(r"^void ", "", r["Kernel"])
        fam = next((f for f in FAMILIES if name.startswith(f)), None)
        if fam is None:
            continue
        a = agg.setdefault(fam, [0, 0.0])
        a[0] += int(r["Dispatches"])
        a[1] += float(r["SumValue"])
    if not agg:
        raise SystemExit(f"{src}: no FETCH_SIZE rows for the engine's kernels")
    kernels = {f: {"dispatches": n, "hbm_bytes_per_launch": int(s / n * 1024 * 2), "raw_fetch_size_kb_per_launch": round(s / n, 1)}
               for f, (n, s) in agg.items()}
    if bench:                       # algorithmic bytes per launch of the same families, from the bench line of the same build
        b = json.load(open(bench))
        if (b.get("extra") or {}).get("libvvhip_build_id") not in (None, vbuild.binary_id()):
            raise SystemExit(f"{bench} was measured on build {b['extra']['libvvhip_build_id']}, not {vbuild.binary_id()}")
        roof = b.get("roofline") or {}
        for fam, ent in ((roof.get("kernel", "").split(" ")[0], roof), ("vv_attn_fused_kernel", roof.get("attention") or {}),
                         ("vv_gemv_kernel", roof.get("gemv_other") or {})):
            if fam in kernels and ent.get("bytes_per_launch"):
                kernels[fam]["algorithmic_bytes_per_launch"] = ent["bytes_per_launch"]
                kernels[fam]["traffic_over_algorithmic"] = round(kernels[fam]["hbm_bytes_per_launch"] / ent["bytes_per_launch"], 3)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = {"libvvhip_build_id": vbuild.binary_id(), "kernels": kernels,
                "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (own pass); KB x 1024 x 2 (gfx950: 64 B counted per 128-B request of a wide "
                        "coalesced stream, MI355X_MICROARCH.md HBM section).  Means are over every dispatch of the run: the decode steps plus the "
                        "prompt-prefill / warm-up launches of the same kernel family (a few % of the dispatches).  The attention unit of a "
                        "decode layer = vv_attn_fused_kernel + vv_attn_merge2_kernel.",
                "source": os.path.basename(src)}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc[key], indent=1))


if __name__ == "__main__":
    main()
