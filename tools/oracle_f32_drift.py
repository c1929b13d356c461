"""How far does the oracle ITSELF drift between float32 and float64 over a T_dec = 800 window?  (CPU only.)

The parity tests of the benchmarked windows hold the HIP path to the fp64 oracle.  With N(0, 1/fan_in) weights, ragged
lengths and feedback the 800-deep recurrence amplifies rounding, so the tolerance a float32 implementation can meet is a
property of the map, not of the kernels: this prints the error of the oracle evaluated in float32 (same equations,
torch-CPU) against itself in float64 -- the yardstick tests/test_gpu_fullshape.py quotes next to the HIP error.

    python tools/oracle_f32_drift.py cfg2v|cfg4 [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import parrot_ref as R
from tests.util import rel_err, rel_err_elem
from tests.test_gpu_fullshape import variant_batch, VARIANTS

which = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 800
kw, init_kw, kb, B, U, ragged = VARIANTS[which]
cfg = R.default_config(**kw)
res = {}
for dt in (torch.float64, torch.float32):
    p = R.init_params(cfg, seed=1234, **init_kw)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(kb)
    p = {k: v.to(dt).requires_grad_() for k, v in p.items()}
    feat, fm, lab, lm = variant_batch(cfg, T, B, U, ragged, seed=77)
    t0 = time.time()
    rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat.to(dt), fm.to(dt), lab, lm.to(dt), None, chunk=100)
    print(which, dt, "T", T, f"{time.time() - t0:.1f} s  cost {float(rc):.8f}", flush=True)
    res[dt] = (float(rc), [x.detach().double() for x in rav], {k: v.grad.detach().double() for k, v in p.items() if v.grad is not None})
c64, av64, g64 = res[torch.float64]
c32, av32, g32 = res[torch.float32]
print("cost rel", abs(c32 - c64) / abs(c64))
for i, n in ((0, "frames"), (1, "kappa"), (2, "w"), (4, "phi")):
    print(n, "norm-wise %.2e element-wise %.2e" % (rel_err(av32[i], av64[i]), rel_err_elem(av32[i], av64[i])))
worst = max(((rel_err(g32[k], g64[k]), k) for k in g64 if float(g64[k].abs().max()) > 1e-12))
print("worst gradient %.2e %s" % worst)
print("kappa end min/mean", float(av64[1][-1].min()), float(av64[1][-1].mean()))
