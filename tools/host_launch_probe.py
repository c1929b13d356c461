"""How long does the HOST spend inside parrot_decoder_seq_fwd / seq_bwd (graph replays), and how long does the device
take?  Development probe for the strand schedule: `python tools/host_launch_probe.py` under PARROT_STRANDS / PARROT_QPART /
PARROT_STRAND_THREADS settings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parrot_amd import _lib, ops
from parrot_amd.model import Parrot

dev = torch.device("cuda:0")
# PROBE_T=100 PROBE_STEPS=1 PROBE_SCANS=0: the short variant for `rocprofv3 --pmc` passes (counter collection costs
# ~30 ms per launch: the full probe's ~20 000 launches do not finish in ten minutes; per-launch traffic does not depend on T)
T, B, U = int(os.environ.get("PROBE_T", "800")), 64, 200
STEPS, SCANS = int(os.environ.get("PROBE_STEPS", "2")), int(os.environ.get("PROBE_SCANS", "3"))
m = Parrot(device=dev, num_layers=2, rnn_h_dim=1024, readouts_dim=1024, encoder_type='bidirectional',
           use_graph=True).initialize()
with torch.no_grad():
    m.get_parameter_dict()['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
g = torch.Generator().manual_seed(1234)
feat = torch.randn(T + 1, B, 63, generator=g).to(dev)
fm = torch.ones(T + 1, B, device=dev)
lab = torch.randint(0, 43, (B, U), generator=g).to(dev)
lm = torch.ones(B, U, device=dev)
for _ in range(STEPS):
    m.zero_grad()
    c, _, _, _ = m.compute_cost(feat, fm, lab, lm, None, 1, B)
    c.backward()
torch.cuda.synchronize()
ws = next(iter(m._train_ws.values()))
plan = ws['plan']
for which, name in ((0, 'parrot_decoder_seq_fwd'), (1, 'parrot_decoder_seq_bwd')):
    for rep in range(SCANS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.call(name, plan, ops._stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host {1e3 * (t1 - t0):7.2f} ms   until the device is done {1e3 * (t2 - t0):7.2f} ms", flush=True)
m.close()
