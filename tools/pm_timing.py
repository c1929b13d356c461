"""Phase timers of the persistent forward scan (PARROT_SCHEDULE=4) at the bench shape: per slot, the time the
workgroups spend in their units and at the barrier, per tick [us]."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PARROT_SCHEDULE", "4")
from parrot_amd.model import Parrot  # noqa: E402

T, B, U, H, L = 800, 64, 200, 1024, int(os.environ.get("L", "2"))
dev = torch.device("cuda:0")
m = Parrot(device=dev, use_graph=True, seed=1234, num_layers=L, rnn_h_dim=H, readouts_dim=H,
           encoder_type='bidirectional').initialize()
g = torch.Generator().manual_seed(1234)
feat = torch.randn(T + 1, B, 63, generator=g).to(dev)
fm = torch.ones(T + 1, B, device=dev)
lab = torch.randint(0, 43, (B, U), generator=g).to(dev)
lm = torch.ones(B, U, device=dev)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        c, upd, av, _ = m.compute_cost(feat, fm, lab, lm, None, 1, B)
    torch.cuda.synchronize()
    print(f"forward {1e3 * (time.perf_counter() - t0):.2f} ms cost {float(c):.5f}")
ws = m._train_ws.get(('dec', T, B, U))
raw = ws['persist_ws'][1024:1024 + 4096].view(torch.int64).cpu().reshape(256, 8).double()
ticks = T + 2 * (L - 1)
rawi = ws['persist_ws'][1024:1024 + 4096].view(torch.int64).cpu().reshape(256, 8)
bad = [w for w in range(256) if rawi[w, 0] > 10**12 or rawi[w, 0] < 0]
print('workgroups with implausible timers:', len(bad), bad[:8], [hex(int(x)) for x in rawi[bad[0]][:6]] if bad else '')
good = [w for w in range(256) if w not in bad]
raw = raw[good]
for s in range(3):
    work, wait = raw[:, s] / 100.0 / ticks, raw[:, 3 + s] / 100.0 / ticks
    print(f"slot {s}: work mean {work.mean():.2f} max {work.max():.2f} min {work.min():.2f} us/tick | "
          f"barrier wait mean {wait.mean():.2f} min {wait.min():.2f} max {wait.max():.2f}")
    srt = torch.sort(work).values
    print("   work by workgroup quartiles:", [round(float(srt[int(i * (len(srt) - 1) / 4)]), 2) for i in range(5)])
tot = (raw[:, :6].sum(1) / 100.0 / ticks)
print(f"per tick total: {tot.mean():.2f} us -> window {tot.mean() * ticks / 1000:.2f} ms")
for s in range(3):
    w = raw[:, s] / 100.0 / ticks
    full = torch.full((256,), float('nan'), dtype=torch.float64)
    full[good] = w
    print(f"slot {s} work per workgroup group: " + " ".join(f"{float(torch.nanmean(full[i:i + 32])):.1f}" for i in range(0, 256, 32)))
raw2 = (ws['persist_ws'][1024 + 4096:1024 + 4096 + 2048].view(torch.int64).cpu().reshape(256, 4).double() / 100.0 / ticks)[good]
for q, n in enumerate(('setup+operand prefetch', 'K loop', 'reduce+sync', 'epilogue+store drain')):
    srt = torch.sort(raw2[:, q]).values
    print(f'gemm stage {n}: quartiles per tick (all units of the workgroup): ' + str([round(float(srt[int(i * (len(srt) - 1) / 4)]), 2) for i in range(5)]))
m.close()
