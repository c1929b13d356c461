import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import ops
dev = torch.device("cuda:0")
def bench(name, a, b, out=None, acc=False, n=10):
    for _ in range(2): ops.gemm(a, b, out=out, accumulate=acc)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): ops.gemm(a, b, out=out, accumulate=acc)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    M, K = a.shape; N = b.shape[1]
    print(f"{name:34s} M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2*M*N*K/dt*1e-12:6.1f} TF", flush=True)
R = 51200
x = torch.randn(R, 1024, device=dev); dg = torch.randn(R, 2048, device=dev); dc = torch.randn(R, 1024, device=dev)
w = torch.randn(1024, 1024, device=dev); w2 = torch.randn(1024, 2048, device=dev); xe = torch.randn(R, 256, device=dev)
bench("TN dWg  x^T dG (acc)", x.t(), dg, out=torch.zeros(1024, 2048, device=dev), acc=True)
bench("TN dWc  x^T dC (acc)", x.t(), dc, out=torch.zeros(1024, 1024, device=dev), acc=True)
bench("TN dWgw w^T dG (acc)", xe.t(), dg, out=torch.zeros(256, 2048, device=dev), acc=True)
bench("NN readout x W", x, w)
bench("NT dx = dy W^T", dg, w2.t())
bench("NN 4096^3", torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev))
