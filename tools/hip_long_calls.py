"""Print the HIP API calls of a `rocprofv3 --hip-trace` CSV that took longer than a threshold (default 2 ms), in time order.
usage: python tools/hip_long_calls.py <dir with *_hip_api_trace.csv> [ms]"""
import csv, glob, sys
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    print(f, len(rows), "calls")
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if d >= thr:
            print(f"  t={(int(r['Start_Timestamp']) - t0) / 1e6:10.1f} ms  {d / 1e6:8.2f} ms  {r['Function']}")
