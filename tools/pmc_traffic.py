"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes)
into profiles/rNN_pmc_traffic.json: HBM-side bytes per launch of the recurrent-step kernel family.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [session label]

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide
coalesced reads, so it is doubled; both counters are in KiB."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter:
                continue
            k = r['Kernel_Name']
            tot[k] += float(r['Counter_Value'])
            cnt[k] += 1
    return tot, cnt


ft, fc = per_kernel(sys.argv[1], 'FETCH_SIZE')
wt, wc = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {"recipe": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernels": {}}
sk_bytes, sk_n = 0.0, 0
pl_bytes, pl_n = 0.0, 0  # the plain step-GEMM launches alone (no attention / state row blocks in the grid)
for k in sorted(ft, key=lambda k: -ft[k]):
    n = fc[k]
    fetch = 2.0 * ft[k] * 1024 / n
    write = (wt.get(k, 0.0) * 1024 / wc[k]) if wc.get(k) else 0.0
    out["kernels"][k[:80]] = {"launches": n, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write)}
    if any(f in k for f in ('sk_kernel', 'ska_kernel', 'skb_kernel', 'wk_kernel', 'wka_kernel', 'wkb_kernel')):  # the step-kernel family
        sk_bytes += (fetch + write) * n
        sk_n += n
        if 'sk_kernel<' in k or k.startswith('wk_kernel('):
            pl_bytes += (fetch + write) * n
            pl_n += n
out["hbm_bytes_per_launch"] = round(sk_bytes / sk_n) if sk_n else None
out["hbm_bytes_per_plain_launch"] = round(pl_bytes / pl_n) if pl_n else None
out["plain_launches"] = pl_n
# digest of the kernel sources the passes ran on: bench.py refuses the figure once they have changed
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_digest  # noqa: E402  (one definition of what the figure is valid for)
out["source_digest"] = kernel_source_digest(root)
if len(sys.argv) > 4:
    out["session"] = sys.argv[4]
out["sk_launches"] = sk_n
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "sk_launches")}))
