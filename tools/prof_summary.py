"""Condense rocprofv3 CSV output (gpurun_out/<dir>) into the small summaries committed under profiles/.
usage: python tools/prof_summary.py stats <kernel_stats.csv> <out.csv> | pmc <counter_collection.csv> <out.csv>"""
import collections
import csv
import sys


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e3:.1f}", r["Percentage"]])


def pmc(src, dst):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(src)):
        k = (r["Kernel_Name"][:100], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
        for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, n, f"{v:.6g}", f"{v / n:.6g}"])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
