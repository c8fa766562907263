#!/usr/bin/env python
"""Summarise a rocprofv3 output database (ROCm 7.2 writes a rocpd SQLite file, <prefix>_results.db) into the CSV files kept
under profiles/:

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db profiles/r02_7b_northstar        # -> *_kernel_stats.csv
    python tools/rocprof_summary.py gpurun_out/pmc/x_results.db profiles/r02_7b_pmc --pmc         # -> *_pmc_by_kernel.csv

--kernel-trace runs: per kernel name: calls, total / average / min / max duration, share of GPU kernel time.
--pmc runs: per kernel name and counter: dispatches, mean and sum of the counter, mean dispatch duration (counter collection
serialises dispatches, so these durations are NOT the ones of the timed run).
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name


def kernel_stats(db, out):
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([short(r[0]), r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
    # the same, split by launch geometry (one line per kernel name x grid): separates the shapes a kernel runs at
    shp = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(end-start), avg(end-start) "
                      "from kernels group by name, grid_x, grid_y, grid_z, workgroup_x order by 7 desc").fetchall()
    with open(out + "_kernel_shapes.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridX", "GridY", "GridZ", "WorkgroupX", "Calls", "TotalDurationNs", "AverageNs"])
        for r in shp[:400]:
            w.writerow([short(r[0])[:80], r[1], r[2], r[3], r[4], r[5], int(r[6]), round(r[7], 1)])
    return rows, tot


CATS = [("gemv", r"vv_gemv_kernel"), ("gemv16p (batch decode)", r"vv_gemv16p_kernel|vv_pack16"), ("attention", r"vv_attn_"), ("tokenizer blocks / convs", r"vv_block1d|vv_normdw|vv_stem_conv|vv_head_conv|vv_shift|vv_dwconv|vv_rmsnorm_rows|vv_gemm_tile|vv_affine"),
        ("prefill gemm", r"vv_gemm3|vv_pack_rows|vv_rope_append"), ("torch / copies", r"at::|rocclr|Cijk|elementwise")]


def timeline(db, out):
    """Decode-phase occupancy of the GPU timeline.  Dispatches are grouped into bursts (a pause of more than 300 us between
    dispatches = the host is between bench phases, not inside a step); a prompt-prefill kernel also ends a burst and is not counted;
    inside the (decode) bursts: busy = union of the kernel intervals, gaps = the rest (kernel boundaries, launch
    latency)."""
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        return
    # every dispatch of the run is a candidate: a burst counts as decode when it holds no prompt-prefill kernel (the bench runs
    # several generate() calls -- timed run, first-audio trials, roofline window -- each with its own prefill; cutting the
    # trace at the LAST prefill kernel, as this function used to, dropped all but the last decode phase)
    win = list(rows)
    if len(win) < 10:
        return
    span = busy = 0
    n_bursts = 0
    b_start = cur_s = cur_e = None
    kept = []
    burst = []

    # kernels that only run outside the step loop: prompt prefill, weight upload / re-packing, the bench's KV fill (torch.randn +
    # vv_kv_import).  Each ends the burst before it and is not counted.
    PREFILL = ("vv_gemm4", "vv_gemm3", "vv_attn_prefill", "vv_pack_rows", "vv_rope_append", "vv_pack_kernel", "vv_kv_import",
               "distribution_elementwise", "vv_cvt_kernel")

    def close(burst):
        nonlocal span, busy, n_bursts
        if len(burst) < 50:          # a stray launch between phases
            return
        # a DECODE burst is made of the step loop's kernels; weight upload / re-packing, KV fills and the oracle's eager legs
        # (memsets, copies, torch kernels back to back) are bursts too and used to be counted as idle decode time
        if sum(1 for n_, _, _ in burst if "vv_gemv" in n_ or "vv_attn_fused" in n_) < 0.3 * len(burst):
            return
        n_bursts += 1
        kept.extend(burst)
        span += max(e for _, _, e in burst) - burst[0][1]
        cs, ce = burst[0][1], burst[0][2]
        for _, s, e in burst[1:]:
            if s > ce:
                busy += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        busy += ce - cs
    last_end = None
    for rec in win:
        if any(p in rec[0] for p in PREFILL):            # a prompt-prefill kernel ends the decode burst before it and is not counted
            close(burst)
            burst = []
            last_end = None
            continue
        if last_end is not None and rec[1] - last_end > 300000:
            close(burst)
            burst = []
        burst.append(rec)
        last_end = max(last_end or 0, rec[2])
    close(burst)
    if not span:
        return
    by = {c: [0, 0] for c, _ in CATS}
    by["other"] = [0, 0]
    for n, s, e in kept:
        for c, pat in CATS:
            if re.search(pat, n):
                by[c][0] += e - s; by[c][1] += 1
                break
        else:
            by["other"][0] += e - s; by["other"][1] += 1
    with open(out + "_timeline.txt", "w") as f:
        f.write(f"decode phase: {n_bursts} bursts, {span / 1e6:.2f} ms, {len(kept)} dispatches: GPU busy (union of kernel intervals) "
                f"{busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %, gaps {100.0 * (span - busy) / span:.1f} %\n")
        f.write("(rocprofv3 --kernel-trace adds ~1 us per dispatch; category sums may exceed the busy time where graph branches overlap)\n")
        for c, (ns, k) in by.items():
            f.write(f"  {c:28s} {k:8d} dispatches {ns / 1e6:10.2f} ms  {100.0 * ns / span:5.1f} % of the bursts\n")


def decode_step_table(db, out, layers_hint=None):
    """Per-kernel time of ONE decode step: the dispatches of the decode bursts (same burst rule as timeline()), grouped by kernel name
    and launch grid, divided by the number of steps in those bursts (= decode-attention dispatches / LM layers; the layer count is
    taken from the most common number of attention launches between two token-embedding / logits launches when not given).
    Writes <out>_decode_step.csv: us per step, launches per step, mean us, kernel, grid."""
    cur = db.cursor()
    rows = cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    PREFILL = ("vv_gemm4", "vv_gemm3", "vv_attn_prefill", "vv_pack_rows", "vv_rope_append", "vv_pack_kernel", "vv_kv_import",
               "distribution_elementwise", "vv_cvt_kernel")
    kept, burst, last_end = [], [], None

    def close(b):
        if len(b) >= 50 and sum(1 for r in b if "vv_gemv" in r[0] or "vv_attn_fused" in r[0]) >= 0.3 * len(b):
            kept.extend(b)
    for rec in rows:
        if any(p in rec[0] for p in PREFILL):
            close(burst); burst = []; last_end = None
            continue
        if last_end is not None and rec[1] - last_end > 300000:
            close(burst); burst = []
        burst.append(rec)
        last_end = max(last_end or 0, rec[2])
    close(burst)
    if not kept:
        return
    n_attn = sum(1 for r in kept if "vv_attn_fused" in r[0])
    layers = layers_hint or 28
    steps = max(1.0, n_attn / layers)
    agg = {}
    for n, s_, e, gx, gy, wx in kept:
        k = (short(n)[:90], gx // max(1, wx), gy, wx)
        a = agg.setdefault(k, [0, 0])
        a[0] += e - s_; a[1] += 1
    tot = sum(a[0] for a in agg.values())
    with open(out + "_decode_step.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([f"# decode bursts only: {len(kept)} dispatches, {steps:.1f} steps (attention launches / {layers} layers), sum of kernel intervals {tot / steps / 1e3:.1f} us per step"])
        w.writerow(["UsPerStep", "LaunchesPerStep", "MeanUs", "Kernel", "Workgroups", "GridY", "Threads"])
        for k, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            w.writerow([round(ns / steps / 1e3, 2), round(c / steps, 2), round(ns / c / 1e3, 2), k[0], k[1], k[2], k[3]])


def gap_report(db, out, min_burst=50):
    """Where the idle time inside the decode bursts sits: every pause between the end of the latest-ending kernel so far and the
    start of the next one, keyed by (kernel before -> kernel after), summed.  Kernel boundaries inside one hipGraph show as
    ~0 (a kernel's recorded interval runs to its successor's start); what shows here are the seams between graph launches /
    eager launches / copies, and the stalls where the stream waits for the host."""
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    PREFILL = ("vv_gemm4", "vv_gemm3", "vv_attn_prefill", "vv_pack_rows", "vv_rope_append", "vv_pack_kernel", "vv_kv_import",
               "distribution_elementwise", "vv_cvt_kernel")
    gaps = {}
    hist = [0] * 8            # <1, <2, <5, <10, <20, <50, <100, >=100 us
    edges = [1, 2, 5, 10, 20, 50, 100]
    tot_gap = tot_span = 0
    burst_n, burst_s, last_end, last_name = 0, None, None, None
    burst_gemv = 0
    pend = []
    def flush():
        nonlocal tot_gap, tot_span
        if burst_n >= min_burst and burst_gemv >= 0.3 * burst_n:
            for k, g in pend:
                gaps.setdefault(k, [0, 0])
                gaps[k][0] += g; gaps[k][1] += 1
                tot_gap += g
                us = g / 1e3
                hist[sum(1 for e in edges if us >= e)] += 1
            tot_span += last_end - burst_s
    for n, s_, e in rows:
        if any(p in n for p in PREFILL) or (last_end is not None and s_ - last_end > 300000):
            flush()
            pend, burst_n, burst_s, last_end, last_name, burst_gemv = [], 0, None, None, None, 0
            if any(p in n for p in PREFILL):
                continue
        if burst_s is None:
            burst_s = s_
        elif s_ > last_end:
            pend.append(((short(last_name)[:48], short(n)[:48]), s_ - last_end))
        burst_n += 1
        if "vv_gemv" in n or "vv_attn_fused" in n:
            burst_gemv += 1
        if last_end is None or e > last_end:
            last_end, last_name = e, n
    flush()
    with open(out + "_gaps.txt", "w") as f:
        f.write(f"idle inside the decode bursts: {tot_gap / 1e6:.3f} ms of {tot_span / 1e6:.3f} ms = {100.0 * tot_gap / max(1, tot_span):.2f} %\n")
        f.write("gap length histogram (us): " + ", ".join(f"{lbl}: {c}" for lbl, c in zip(["<1", "1-2", "2-5", "5-10", "10-20", "20-50", "50-100", ">=100"], hist)) + "\n")
        f.write("largest contributors (kernel whose end precedes the gap -> kernel that follows): total us, count, mean us\n")
        for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
            f.write(f"  {g / 1e3:10.1f} us  {c:6d} x {g / 1e3 / c:7.2f} us   {k[0]}  ->  {k[1]}\n")


def dump_around(db, out, pattern, occurrence, count=14):
    """per-dispatch listing (start offset, duration, gap after the previous kernel's end) around the `occurrence`-th dispatch
    whose name matches `pattern`: where the time between two kernels of a chain goes"""
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    hits = [i for i, r in enumerate(rows) if re.search(pattern, r[0])]
    if len(hits) <= occurrence:
        return
    i0 = max(0, hits[occurrence] - count // 2)
    with open(out + "_dispatches.txt", "w") as f:
        base = rows[i0][1]
        prev_end = None
        for n, s_, e in rows[i0:i0 + count]:
            gap = (s_ - prev_end) / 1e3 if prev_end is not None else 0.0
            f.write(f"{(s_ - base) / 1e3:10.1f} us  dur {(e - s_) / 1e3:9.1f} us  gap {gap:8.1f} us  {short(n)[:70]}\n")
            prev_end = e


def pmc_stats(db, out):
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(end-start) "
                       "from counters_collection group by kernel_name, counter_name order by 5 desc").fetchall()
    with open(out + "_pmc_by_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "MeanValue", "SumValue", "MeanDispatchNs"])
        for r in rows:
            w.writerow([short(r[0]), r[1], r[2], round(r[3], 2), round(r[4], 1), round(r[5], 1)])
    return rows


if __name__ == "__main__":
    db = sqlite3.connect(sys.argv[1])
    if "--pmc" in sys.argv:
        for r in pmc_stats(db, sys.argv[2])[:30]:
            print(f"{short(r[0])[:90]:90s} {r[1]:32s} n={r[2]:6d} mean={r[3]:14.1f} dur={r[5] / 1e3:9.2f}us")
    else:
        rows, tot = kernel_stats(db, sys.argv[2])
        timeline(db, sys.argv[2])
        gap_report(db, sys.argv[2])
        decode_step_table(db, sys.argv[2], int(sys.argv[sys.argv.index("--layers") + 1]) if "--layers" in sys.argv else None)
        if "--around" in sys.argv:
            k = sys.argv.index("--around")
            dump_around(db, sys.argv[2], sys.argv[k + 1], int(sys.argv[k + 2]))
        print(f"total kernel time {tot / 1e6:.2f} ms")
        for r in rows[:25]:
            print(f"{short(r[0])[:100]:100s} {r[1]:7d} {r[2] / 1e6:9.2f} ms  avg {r[3] / 1e3:9.2f} us  {100 * r[2] / tot:5.1f}%")
