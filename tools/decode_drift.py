"""How does the HIP decode (BASELINE configs[2] shapes: batch 16, H=1024, weak feedback, 1000 frames) drift from the
fp64 oracle over the frames?  Prints the norm-wise error of the frames up to step t for a few t, for the persistent
machine and for the launch path, and the error of a second HIP run against the first (run-to-run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import parrot_ref as R
from parrot_amd.model import Parrot
from tests.util import make_batch

dev = torch.device("cuda:0")
kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024, weak_feedback=True)
cfg = R.default_config(**kw)
p = R.init_params(cfg, seed=29, scale_by_fan_in=True)
p['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.9)
N, U, S = 16, 200, int(os.environ.get("DRIFT_S", "1000"))
_, _, lab, lm, _ = make_batch(cfg, 2, N, U, seed=31)
with torch.no_grad():
    ref = R.sample_model(p, cfg, lab, lm, None, S)
rx = ref[0].double()
res = {}
for mode in ("1", "0"):
    os.environ["PARROT_SAMPLE_PERSIST"] = mode
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    outs = m.sample_model(lab.numpy(), lm.float().numpy(), None, None, N, S)
    x = torch.from_numpy(outs[0]).double()
    res[mode] = x
    line = []
    for t in (10, 30, 60, 100, 200, 300, 500, 700, 1000):
        if t > S:
            break
        e = float((x[:t] - rx[:t]).abs().max() / rx[:t].abs().max())
        line.append(f"t<={t}: {e:.1e}")
    print(f"PARROT_SAMPLE_PERSIST={mode}: " + "  ".join(line), flush=True)
    k = torch.from_numpy(outs[1]).double()
    print(f"   kappa end: hip {float(k[-1].mean()):.4f} oracle {float(ref[1][-1].mean()):.4f}", flush=True)
    m.close()
d = float((res["1"] - res["0"]).abs().max() / rx.abs().max())
print(f"machine vs launch path (both fp32): {d:.1e}")
