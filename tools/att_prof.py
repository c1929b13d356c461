import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import ops
dev = torch.device("cuda:0")
B,H,A,U,E = 64,1024,10,200,256
g = torch.Generator().manual_seed(0)
h1 = torch.randn(B, H, generator=g).to(dev); WT = (torch.randn(3*A, H, generator=g)*0.02).to(dev)
batt = torch.zeros(3*A, device=dev); kp = torch.rand(B, A, generator=g).to(dev)*3
ctx = torch.randn(B, U, E, generator=g).to(dev)
for _ in range(50): ops.gmm_attention_fwd(h1, WT, batt, kp, ctx)
torch.cuda.synchronize()

a,b,k,phi,w = ops.gmm_attention_fwd(h1, WT, batt, kp, ctx)
dw = torch.randn(B, E, device=dev); dk = torch.zeros(B, A, device=dev); dh = torch.zeros(B, H, device=dev)
for _ in range(50): ops.gmm_attention_bwd(dw, ctx, a, b, k, kp, WT, dk, dh)
torch.cuda.synchronize()
