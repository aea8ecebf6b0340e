"""GPU idle time between the kernels of a train step, from a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_analysis.py <p_kernel_trace.csv> [marker kernel substring = adam_multi] -> per step: span, busy (union of kernel intervals), idle,
number of launches, and the kernels that most often START after an idle gap."""
import csv, sys
from collections import Counter, defaultdict
path = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else "adam_multi"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step ends with the last marker kernel of a burst (markers closer than 2 ms belong to one step)
ends = [i for i, r in enumerate(rows) if marker in r[2]]
bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or rows[ends[k + 1]][0] - rows[i][1] > 2_000_000]
print("steps found:", len(bounds))
for a, b in list(zip(bounds, bounds[1:]))[-4:]:
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    gaps = defaultdict(int); gcount = Counter()
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps[n.split("(")[0][:60]] += s - cur_e; gcount[n.split("(")[0][:60]] += 1; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in seg)
    print(f"step: span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f}  idle {(span - busy) / 1e6:.3f}  kernel-time sum {ksum / 1e6:.3f}  launches {len(seg)}  mean gap {((span - busy) / max(1, sum(gcount.values()))) / 1e3:.2f} us over {sum(gcount.values())} gaps")
top = sorted(gaps.items(), key=lambda kv: -kv[1])[:12]
print("idle in front of (last step):")
for n, g in top: print(f"  {g / 1e3:8.1f} us  x{gcount[n]:4d}  {n}")
