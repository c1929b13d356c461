"""Phase timers of the persistent phase machine: per slot, the time the workgroups spend in their units and at the
barrier, per tick [us].  MODE=train (default): forward scan of the bench shape (PARROT_SCHEDULE=4); MODE=decode:
BASELINE configs[2] (batch 16, H=1024, weak feedback)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODE = os.environ.get("MODE", "train")
os.environ.setdefault("PARROT_SCHEDULE", "4")
from parrot_amd.model import Parrot  # noqa: E402

dev = torch.device("cuda:0")
H, L = 1024, int(os.environ.get("L", "2"))
g = torch.Generator().manual_seed(1234)
if MODE == "train":
    T, B, U = 800, 64, 200
    m = Parrot(device=dev, use_graph=True, seed=1234, num_layers=L, rnn_h_dim=H, readouts_dim=H,
               encoder_type='bidirectional').initialize()
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
    pw = m._train_ws.get(('dec', T, B, U))['persist_ws']
    ticks, nslots = T + 2 * (L - 1), 3
else:
    N, U, S = 16, 100, 1000
    m = Parrot(device=dev, use_graph=True, seed=1234, num_layers=L, rnn_h_dim=H, readouts_dim=H,
               encoder_type='bidirectional', weak_feedback=True).initialize()
    lab = torch.randint(0, 43, (N, U), generator=g)
    lm = torch.ones(N, U)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = m.sample_model_device(lab, lm, None, N, S)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"decode {1e3 * dt:.2f} ms = {1e6 * dt / S:.2f} us per step")
    pw = m._sample_ws.get((S, N, U))['pm']['ws']
    pieces = os.environ.get("PARROT_PM_PIECES", "1") != "0"  # plans_decode.hip build_persist_pieces: 2L + 2 phases, S + 1 ticks
    from parrot_amd import _lib as _plib
    kind = int(_plib.load().parrot_sample_is_persistent(m._sample_ws.get((S, N, U))['plan']))
    ticks, nslots = {3: (S + 2, 2 * L + 1), 2: (S + 1, 2 * L + 2)}.get(kind, (S, 2 * L + 3))
    print(f"plan kind {kind}: {nslots} phases per step")
rawi = pw[1024:1024 + 256 * 48].view(torch.int64).cpu().reshape(256, 24)
bad = [w for w in range(256) if rawi[w, 0] > 10**12 or rawi[w, 0] < 0]
print('workgroups with implausible timers (XCC whose s_memrealtime does not tick):', len(bad))
good = [w for w in range(256) if w not in bad]
raw = rawi.double()[good] / 100.0 / ticks
tot = 0.0
for s in range(nslots):
    work, wait = raw[:, s], raw[:, 9 + s]
    srt = torch.sort(work).values
    q = [round(float(srt[int(i * (len(srt) - 1) / 4)]), 2) for i in range(5)]
    print(f"slot {s}: work quartiles {q} us/tick | barrier wait mean {wait.mean():.2f} min {wait.min():.2f}")
    tot += float((work + wait).mean())
print(f"per tick: {tot:.2f} us -> {tot * ticks / 1000:.2f} ms")
for q, n in enumerate(('setup+operand prefetch', 'K loop', 'reduce+sync', 'epilogue+store drain')):
    srt = torch.sort(raw[:, 18 + q]).values
    print(f'gemm stage {n}: quartiles per tick over the workgroups: ' +
          str([round(float(srt[int(i * (len(srt) - 1) / 4)]), 2) for i in range(5)]))
m.close()
