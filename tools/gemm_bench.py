"""TF/s of the batched GEMM on the shapes of one cfg2 training step (deferred weight gradients: TN products with
K = T*B = 51200; readout / feedback projections: NN and NT), weighted by how often each runs in a step, in the two
f32-grade modes of the library side by side: the f32-input MFMA kernel (bg_kernel8) and the split-bf16 kernel
(bgs_kernel, three bf16 terms per operand, six bf16 MFMAs per block: PARROT_PRECISION_BF16X3).

    python tools/gemm_bench.py

Every result is checked against a float64 product of a 64-row sample of the output: `err` = max |c - ref| / max |ref|,
`rms` = rms(c - ref) / rms(ref)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import ops
dev = torch.device("cuda:0")
MODES = (("f32", ops.PRECISION_F32), ("x3 ", ops.PRECISION_BF16X3))
tot = {m: [0.0, 0.0] for m, _ in MODES}
def bench(name, a, b, out=None, acc=False, n=6, weight=1):
    M, K = a.shape; N = b.shape[1]
    g = torch.Generator().manual_seed(0)
    rows = torch.randint(0, M, (64,), generator=g).to(dev)
    ref = a[rows].double() @ b.double()
    for mode, prec in MODES:
        with ops.gemm_precision(prec):
            if out is not None:
                out.zero_()
            r = ops.gemm(a, b, out=out, accumulate=acc)
            d = r[rows].double() - ref
            err = float(d.abs().max() / ref.abs().max())
            rms = float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
            ops.gemm(a, b, out=out, accumulate=acc)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(n): ops.gemm(a, b, out=out, accumulate=acc)
            torch.cuda.synchronize(); dt = (time.time() - t0) / n
        tot[mode][0] += dt * weight; tot[mode][1] += 2.0 * M * N * K * weight
        print(f"{mode} {name:26s} M={M:5d} N={N:5d} K={K:5d}: {dt*1e6:8.1f} us {2*M*N*K/dt*1e-12:6.1f} TF  err {err:.1e} rms {rms:.1e}", flush=True)
R = int(os.environ.get("GEMM_ROWS", "51200"))
x = torch.randn(R, 1024, device=dev); dg = torch.randn(R, 2048, device=dev); dc = torch.randn(R, 1024, device=dev)
w = torch.randn(1024, 1024, device=dev); w2 = torch.randn(1024, 2048, device=dev); xe = torch.randn(R, 256, device=dev)
w63 = torch.randn(1024, 63, device=dev); d63 = torch.randn(R, 63, device=dev)
# deferred weight gradients of the cfg2 scan (model.py _weight_grad_rows): per layer h^T dG, rh^T dC, w^T dG, w^T dC, h_j^T d*
bench("TN dWg  h^T dG (acc)", x.t(), dg, out=torch.zeros(1024, 2048, device=dev), acc=True, weight=3)
bench("TN dWc  h^T dC (acc)", x.t(), dc, out=torch.zeros(1024, 1024, device=dev), acc=True, weight=3)
bench("TN dWgw w^T dG (acc)", xe.t(), dg, out=torch.zeros(256, 2048, device=dev), acc=True, weight=2)
bench("TN dWcw w^T dC (acc)", xe.t(), dc, out=torch.zeros(256, 1024, device=dev), acc=True, weight=2)
bench("TN dWo  r^T dy63", x.t(), d63, weight=1)
bench("NN readout h W", x, w, weight=3)
bench("NN out r W63", x, w63, weight=1)
bench("NT dh = dr W^T", x, w.t(), weight=3)
bench("NT dr = dy63 W63^T", d63, w63.t(), weight=1)
for m, _ in MODES:
    print(f"{m} weighted total {tot[m][0]*1e3:.2f} ms  {tot[m][1]/tot[m][0]*1e-12:.1f} TF")
# exactness of the split on hard operands: wide dynamic range, same sign (no cancellation to hide behind)
a = (torch.rand(512, 8192, device=dev) + 0.5) * torch.exp2(torch.randint(-12, 12, (512, 8192), device=dev).float())
b = (torch.rand(8192, 512, device=dev) + 0.5)
ref = a.double() @ b.double()
for mode, prec in MODES:
    with ops.gemm_precision(prec):
        r = ops.gemm(a, b)
    rel = ((r.double() - ref).abs() / ref.abs()).max()
    print(f"{mode} same-sign wide-range operands 512x8192x512: max element-wise relative error {float(rel):.2e}")
