"""Per launch TYPE of a kernel trace: average duration by (kernel, grid size) -- tells the gate / candidate / attention launches
of the forward tick and the S' / X / Y launches of the backward tick apart (they share three kernel names).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py ... ;  python tools/launch_types.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    if "sk" not in name and "wk" not in name:
        continue
    key = (name[:48], int(r.get("Grid_Size_X", 0)) // max(1, int(r.get("Workgroup_Size_X", 1))), int(r.get("Grid_Size_Y", 1)),
           int(r.get("Grid_Size_Z", 1)))
    a = acc[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
tot = sum(a[1] for a in acc.values())
print(f"{'kernel':50s} {'wg x':>6s} {'y':>3s} {'z':>3s} {'launches':>9s} {'avg us':>8s} {'share':>6s}")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if a[0] < 20:
        continue
    print(f"{k[0]:50s} {k[1]:6d} {k[2]:3d} {k[3]:3d} {a[0]:9d} {a[1] / a[0]:8.2f} {100 * a[1] / tot:5.1f}%")
