"""TF/s of the batched f32 GEMM on the shapes of one cfg2 training step (deferred weight gradients: TN products with
K = T*B = 51200; readout / feedback projections: NN and NT), weighted by how often each runs in a step.

    python tools/gemm_bench.py

Every result is checked against a float64 product of a 64-row sample of the output.  Round 4 ran it once per macro-tile
variant of the kernel (PARROT_GEMM_VARIANT=0..6 of a build that has since been removed: profiles/r04_gemm_tile_variants.txt);
the tag printed in front of every line is that variable (v0 = the 128 x 128 x 16 kernel of the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import ops
dev = torch.device("cuda:0")
var = os.environ.get("PARROT_GEMM_VARIANT", "0")
tot_t = tot_f = 0.0
def bench(name, a, b, out=None, acc=False, n=6, weight=1):
    global tot_t, tot_f
    if out is not None:
        out.zero_()
    r = ops.gemm(a, b, out=out, accumulate=acc)
    g = torch.Generator().manual_seed(0)
    rows = torch.randint(0, a.shape[0], (64,), generator=g).to(dev)
    ref = a[rows].double() @ b.double()
    err = float((r[rows].double() - ref).abs().max() / ref.abs().max())
    ops.gemm(a, b, out=out, accumulate=acc)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): ops.gemm(a, b, out=out, accumulate=acc)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    M, K = a.shape; N = b.shape[1]
    tot_t += dt * weight; tot_f += 2.0 * M * N * K * weight
    print(f"v{var} {name:30s} M={M:5d} N={N:5d} K={K:5d}: {dt*1e6:8.1f} us {2*M*N*K/dt*1e-12:6.1f} TF  err {err:.1e}", flush=True)
R = 51200
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
print(f"v{var} weighted total {tot_t*1e3:.2f} ms  {tot_f/tot_t*1e-12:.1f} TF")
if var == "0":
    bench("NN 4096^3", torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev))
