"""Effective shader clock of the batched f32 GEMM under load (DVFS): run under
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python tools/gemm_clock.py
and reduce with `python tools/gemm_clock.py --reduce <dir>`: clock = GRBM_GUI_ACTIVE / kernel duration.  The f32 MFMA
peak of 157.3 TFLOP/s assumes 2.4 GHz; what the kernel can reach is 256 CUs x 4 SIMDs x 64 flop/clk x the clock it
actually sustains on random operands."""
import csv
import glob
import os
import sys

if "--reduce" in sys.argv:
    root = sys.argv[sys.argv.index("--reduce") + 1]
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    out = {}
    for r in rows:
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE" or "bg_kernel" not in r["Kernel_Name"] and "bgx" not in r["Kernel_Name"]:
            continue
        t0, t1 = float(r.get("Start_Timestamp", 0)), float(r.get("End_Timestamp", 0))
        if t1 <= t0:
            continue
        k = r["Kernel_Name"][:60]
        out.setdefault(k, []).append((float(r["Counter_Value"]), t1 - t0))
    for k, v in out.items():
        v = [x for x in v if x[1] > 2e5]  # the long (K = 51200) products only
        if not v:
            continue
        cyc = sum(x[0] for x in v) / len(v)
        ns = sum(x[1] for x in v) / len(v)
        # GRBM_GUI_ACTIVE is summed over the chip's XCDs by the tool on some versions: report both readings
        print(f"{k}: {len(v)} launches, avg {ns * 1e-3:.1f} us, GRBM_GUI_ACTIVE {cyc:.0f} -> {cyc / ns:.3f} GHz "
              f"(per-XCD reading {cyc / ns / 8:.3f} GHz); f32 MFMA peak at that clock "
              f"{256 * 4 * 64 * cyc / ns * 1e-3:.1f} / {256 * 4 * 64 * cyc / ns / 8 * 1e-3:.1f} TFLOP/s")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import ops
dev = torch.device("cuda:0")
R = 51200
x = torch.randn(R, 1024, device=dev); dg = torch.randn(R, 2048, device=dev)
out = torch.zeros(1024, 2048, device=dev)
for _ in range(8):
    ops.gemm(x.t(), dg, out=out, accumulate=True)
torch.cuda.synchronize()
print("done")
