"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv (development aid)."""
import csv
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(float)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name'][:50]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in tot:
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:32s} {tot[k][c] / cnt[k][c]:16.1f}  (x{cnt[k][c]})")
