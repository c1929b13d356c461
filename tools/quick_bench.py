"""Quick timing probe of the cfg2 training step (development aid, not the driver's bench.py)."""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parrot_amd import ops
from parrot_amd.model import Parrot

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=800)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--U", type=int, default=200)
ap.add_argument("--H", type=int, default=1024)
ap.add_argument("--L", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--cell", default="gru")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Parrot(device=dev, num_layers=a.L, rnn_h_dim=a.H, readouts_dim=a.H, encoder_type='bidirectional',
           use_graph=bool(a.graph), cell_type=a.cell).initialize()
g = torch.Generator().manual_seed(1234)
feat = torch.randn(a.T + 1, a.B, 63, generator=g).to(dev)
fm = torch.ones(a.T + 1, a.B, device=dev)
lab = torch.randint(0, 43, (a.B, a.U), generator=g).to(dev)
lm = torch.ones(a.B, a.U, device=dev)
mom = torch.zeros_like(m.flat_parameters)
var = torch.zeros_like(m.flat_parameters)
for s in range(a.steps + 1):
    torch.cuda.synchronize()
    t0 = time.time()
    m.zero_grad()
    cost, upd, av, _ = m.compute_cost(feat, fm, lab, lm, None, 1, a.B)
    torch.cuda.synchronize()
    t1 = time.time()
    cost.backward()
    torch.cuda.synchronize()
    t2 = time.time()
    nrm = ops.sumsq(m.flat_gradients)
    ops.adam_clip_step(m.flat_parameters, m.flat_gradients, mom, var, nrm, s + 1)
    torch.cuda.synchronize()
    t3 = time.time()
    print(f"step {s}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.2f} ms  "
          f"cost {float(cost):.5f}  frames/s {a.B*a.T/(t3-t0):.0f}", flush=True)
