"""Short probe for `rocprofv3 --pmc` passes (counter collection costs tens of ms per launch, so the probe must stay
near 1 000 launches): ONE training step of a configuration at a short window, eager launches (no hipGraph).

    PROBE_CFG=cfg2|cfg4  PROBE_T=100  python tools/pmc_probe.py

Per-launch traffic of the step kernels does not depend on T (the same launches, fewer of them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parrot_amd.model import Parrot

dev = torch.device("cuda:0")
cfg = os.environ.get("PROBE_CFG", "cfg2")
T, B, U = int(os.environ.get("PROBE_T", "100")), 64, 200
kw = (dict(num_layers=3, rnn_h_dim=1536, readouts_dim=1536, cell_type='lstm', compute_dtype='bf16') if cfg == "cfg4"
      else dict(num_layers=2, rnn_h_dim=1024, readouts_dim=1024))
m = Parrot(device=dev, encoder_type='bidirectional', use_graph=os.environ.get("PROBE_GRAPH", "0") == "1", **kw).initialize()
with torch.no_grad():
    m.get_parameter_dict()['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
g = torch.Generator().manual_seed(1234)
feat = torch.randn(T + 1, B, 63, generator=g).to(dev)
fm = torch.ones(T + 1, B, device=dev)
lab = torch.randint(0, 43, (B, U), generator=g).to(dev)
lm = torch.ones(B, U, device=dev)
for _ in range(int(os.environ.get("PROBE_STEPS", "1"))):
    m.zero_grad()
    c, _, _, _ = m.compute_cost(feat, fm, lab, lm, None, 1, B)
    c.backward()
torch.cuda.synchronize()
print("probe", cfg, "T", T, "cost", float(c))
m.close()
