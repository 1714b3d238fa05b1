"""Turns rocprofv3 CSV output into the small summaries committed under profiles/.
  python tools/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.md
  python tools/summarize_rocprof.py pmc    <dir with *_counter_collection.csv> COUNTER profiles/rNN_pmc_COUNTER.md
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def newest(pattern):
    """the most recent match (gpurun merges every call's output into the same local directory)"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def stats(src, dst):
    f = newest(src + "/**/*_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as o:
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            o.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | "
                    f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |\n")
        o.write(f"\ntotal kernel time {tot/1e6:.2f} ms\n")
    print("wrote", dst)


def pmc(src, counter, dst):
    f = newest(src + "/**/*_counter_collection.csv")
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    out = {}
    with open(dst, "w") as o:
        o.write(f"| kernel | dispatches | {counter} total | per dispatch |\n|---|---|---|---|\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"| `{k[:90]}` | {n} | {v:.0f} | {v/n:.1f} |\n")
            out[k] = {"dispatches": n, "total": v, "per_dispatch": v / n}
    json.dump(out, open(dst.replace(".md", ".json"), "w"), indent=1)
    print("wrote", dst)


def traffic(fetch_dir, write_dir, dst_md, frame_kernel="slow_engine_kernel"):
    """HBM-side bytes per kernel and per decode frame from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KB).
    On gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled here; WRITE_SIZE is
    exact.  A frame = one dispatch of `frame_kernel`."""
    def load(d, counter):
        f = newest(d + "/**/*_counter_collection.csv")
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        return agg
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    frames = max([v[0] for k, v in fe.items() if frame_kernel in k] + [1])
    rows, tot = [], 0.0
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1])):
        nf, fkb = fe.get(k, [0, 0.0])
        nw, wkb = wr.get(k, [0, 0.0])
        b = fkb * 2 * 1024 + wkb * 1024
        rows.append((k, nf, fkb * 2 * 1024 / max(nf, 1), wkb * 1024 / max(nw, 1), b / frames))
        tot += b
    with open(dst_md, "w") as o:
        o.write(f"frames (dispatches of {frame_kernel}): {frames}\n\n| kernel | dispatches | read bytes / dispatch (FETCH_SIZE x 2) | written bytes / dispatch | bytes / frame |\n|---|---|---|---|---|\n")
        for k, n, rb, wb, pf in rows:
            o.write(f"| `{k[:90]}` | {n} | {rb:,.0f} | {wb:,.0f} | {pf:,.0f} |\n")
        o.write(f"\nHBM-side bytes per frame, all kernels of the run (prefill and codec launches included): {tot / frames:,.0f}\n")
    dec = sum(pf for k, n, rb, wb, pf in rows if any(t in k for t in ("engine_kernel", "gemv_kernel", "samp_", "sample_", "attn_", "embed_kernel")))
    json.dump({"frames": frames, "hbm_bytes_per_frame": round(dec), "per_kernel": {k[:90]: {"dispatches": n, "read_bytes_per_dispatch": round(rb),
               "written_bytes_per_dispatch": round(wb), "bytes_per_frame": round(pf)} for k, n, rb, wb, pf in rows}},
              open(dst_md.replace(".md", ".json"), "w"), indent=1)
    print("wrote", dst_md)


def trace(src, dst, first="rvq_gather"):
    """Per-call list of the LAST pass that starts with kernel `first` (one codec decode) from a --kernel-trace run:
    kernel, grid in workgroups, duration."""
    f = newest(src + "/**/*_kernel_trace.csv")
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    rows = rows[idx[-1]:] if idx else rows
    tot = 0.0
    with open(dst, "w") as o:
        o.write("| # | kernel | grid (workgroups) | us |\n|---|---|---|---|\n")
        for i, r in enumerate(rows):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            tot += d
            g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])),
                 int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
            o.write(f"| {i} | `{r['Kernel_Name'].replace('void ft::', '').replace('ft::', '')[:60]}` | {g[0]} x {g[1]} x {g[2]} | {d:.1f} |\n")
        span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
        o.write(f"\nkernel time {tot:.0f} us, first start to last end {span:.0f} us\n")
    print("wrote", dst)


if __name__ == "__main__" and sys.argv[1] != "mfma":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], *(sys.argv[4:5]))
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])


def mfma(src, dst):
    """MFMA utilisation per kernel from one --pmc pass of SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE:
    util = MFMA-busy cycles / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    f = newest(src + "/**/*_counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[r["Kernel_Name"]] += 1
    rows = []
    for k, c in agg.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or act <= 0:
            continue
        cyc = act / 8.0
        rows.append((k, n[k], busy, cyc, busy / (cyc * 1024.0)))
    with open(dst, "w") as o:
        o.write("| kernel | dispatches | MFMA-busy cycles (all SIMDs) | kernel cycles | MFMA utilisation | dense-bf16 TFLOP/s equivalent |\n|---|---|---|---|---|---|\n")
        for k, nn, busy, cyc, u in sorted(rows, key=lambda r: -r[2]):
            o.write(f"| `{k[:80]}` | {nn} | {busy:.3g} | {cyc:.3g} | {100 * u:.1f} % | {2500 * u:.0f} |\n")
    print("wrote", dst)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mfma":
    mfma(sys.argv[2], sys.argv[3])
