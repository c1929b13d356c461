"""Reads a rocprofv3 kernel-trace CSV and reports, for the last `frac` of the trace, the sum of kernel
durations, the union of busy time and the wall span (development aid: do graph branches overlap?)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
n = len(rows)
lo = int(n * float(sys.argv[2]) if len(sys.argv) > 2 else 0)
rows = rows[lo:]
span = max(e for _, e, _ in rows) - rows[0][0]
tot = sum(e - s for s, e, _ in rows)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  sum(dur) {tot/1e6:.2f} ms  union(busy) {busy/1e6:.2f} ms")
agg = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    agg[k][0] += 1
    agg[k][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  {t/1e6:8.2f} ms  {c:6d} x {t/c/1e3:7.2f} us  {k}")
