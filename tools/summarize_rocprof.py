"""Turns rocprofv3 CSV output into the small summaries committed under profiles/.
  python tools/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.md
  python tools/summarize_rocprof.py pmc    <dir with *_counter_collection.csv> COUNTER profiles/rNN_pmc_COUNTER.md
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def stats(src, dst):
    f = glob.glob(src + "/**/*_kernel_stats.csv", recursive=True)[0]
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
    f = glob.glob(src + "/**/*_counter_collection.csv", recursive=True)[0]
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
