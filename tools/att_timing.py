"""Phase timers of the attention row blocks inside the heterogeneous step launches (ska_kernel forward, skb_kernel backward)
of one cfg2 training step, as the replayed graph runs them.  Needs the development build with in-kernel stamps:

    python -m parrot_amd.build --timers && PARROT_HIP_LIB=parrot_amd/libparrot_hip_timers.so python tools/att_timing.py

Every wave of a row block stamps the 100 MHz wall clock at its phase boundaries; the stamps left in the buffer are those of
the LAST forward tick / LAST backward tick of the window.  Printed: per phase, quartiles over the 64 rows' waves, in us after
the earliest entry of any row block of that launch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PARROT_HIP_LIB", os.path.join(ROOT, "parrot_amd", "libparrot_hip_timers.so"))
import torch  # noqa: E402
from parrot_amd import _lib  # noqa: E402
from parrot_amd.model import Parrot  # noqa: E402
from parrot_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
T, B, U, H, L = int(os.environ.get("T", "800")), 64, 200, 1024, 2
m = Parrot(device=dev, use_graph=True, seed=1234, num_layers=L, rnn_h_dim=H, readouts_dim=H, encoder_type='bidirectional').initialize()
with torch.no_grad():
    m.get_parameter_dict()['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
tr = Trainer(m)
g = torch.Generator().manual_seed(1234)
batch = (torch.randn(T + 1, B, 63, generator=g).to(dev), torch.ones(T + 1, B, device=dev),
         torch.randint(0, 43, (B, U), generator=g).to(dev), torch.ones(B, U, device=dev))
for _ in range(3):
    tr.step(*batch, None, 1)
torch.cuda.synchronize()
lib = _lib.load()
sk = torch.zeros(4096 * 16 * 8, dtype=torch.int64, device=dev)
att = torch.zeros(2 * 64 * 16 * 16, dtype=torch.int64, device=dev)
lib.parrot_debug_set_timers.argtypes = [C.c_void_p, C.c_void_p]
rc = lib.parrot_debug_set_timers(C.c_void_p(sk.data_ptr()), C.c_void_p(att.data_ptr()))
assert rc == 0, rc
tr.step(*batch, None, 1)
torch.cuda.synchronize()
a = att.cpu().view(2, 64, 16, 16).double()
names = {0: ["entry", "projection done (wave)", "projection in LDS (barrier)", "window parameters (barrier)", "phi + support (barrier)",
             "context rows multiplied (wave)", "partial sums in LDS (barrier)", "w stored (issued)"],
         1: ["entry", "dw total + window parameters loaded (wave)", "... in LDS (barrier)", "dphi of this wave's rows", "dphi in LDS (barrier)",
             "mixture reductions (barrier)", "dp in LDS (barrier)", "dh1 updated", "layer 0 state backward stored"]}
for d, title in ((0, "attention FORWARD row blocks (ska_kernel, 8 waves per row)"), (1, "attention BACKWARD row blocks (skb_kernel, 16 waves per row)")):
    x = a[d]
    ent = x[:, :, 0]
    live = ent > 0
    if not bool(live.any()):
        print(title, ": no stamps"); continue
    t0 = float(ent[live].min())
    print(f"{title}: {int(live.sum())} waves; us after the launch's earliest row-block entry (min / q1 / median / q3 / max)")
    for i, nm in enumerate(names[d]):
        v = x[:, :, i][live & (x[:, :, i] >= t0) & (x[:, :, i] - t0 < 1e5)]
        if v.numel() == 0:
            continue
        v = ((v - t0) * 0.01).sort().values
        n = v.numel()
        print(f"  [{i}] {nm:46s} n={n:5d} {float(v[0]):6.2f} {float(v[n // 4]):6.2f} {float(v[n // 2]):6.2f} {float(v[3 * n // 4]):6.2f} {float(v[-1]):6.2f}")
