"""Development aid: one SampleRNN training window (cfg5 widths, B = 4 x 320 samples) in both f32-grade GEMM modes; compares
the ReLU masks of the sample-level MLP between the modes (a pre-activation within rounding of 0 flips its mask bit, and with
it a whole term of every upstream gradient: a discontinuity of the function, not an error of a product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parrot_amd import _lib, ops
dev = torch.device("cuda:0")
rec = []
orig_gemm = ops.gemm
def spy(a, b, bias=None, out=None, accumulate=False, act=ops.ACT_NONE, alpha=1.0, split_k=0):
    r = orig_gemm(a, b, bias=bias, out=out, accumulate=accumulate, act=act, alpha=alpha, split_k=split_k)
    if act == ops.ACT_RELU:
        pre = (a.double() @ b.double() + (bias.double() if bias is not None else 0))
        rec.append((r.detach().clone(), pre))
    return r
ops.gemm = spy
from oracle import samplernn_ref as S
from parrot_amd.sampleRNN import lib
from parrot_amd.sampleRNN.models.conditional import three_tier as tt
lib.delete_all_params(); lib.set_device(dev)
tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
c = S.config(); p = S.init_params(c, seed=8, perturb=0.1); lib.set_params(p)
g = torch.Generator().manual_seed(3)
B, S_len = 4, 320
seq = (torch.randn(B, S_len + 80, generator=g) * 12 + 128).round().clamp(0, 255).long()
seq[:, ::9] = 128
feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
mask = torch.ones(B, S_len + 80, dtype=torch.float64)
mask[1, 250:] = 0
mask[3, 333:] = 0
h0 = torch.randn(B, 1, 1024, generator=g, dtype=torch.float64) * 0.3
bh0 = torch.randn(B, 1, 1024, generator=g, dtype=torch.float64) * 0.3
masks = {}
for mode in (ops.PRECISION_F32, ops.PRECISION_BF16X3):
    ops.set_full_precision(mode)
    rec.clear()
    cost, ip_cost, *_ = tt.compute_cost(seq.to(dev), feats.float().to(dev), h0.float().to(dev), bh0.float().to(dev), 1, mask.float().to(dev))
    masks[mode] = [(r > 0, pre) for r, pre in rec]
    print("mode", mode, "cost", float(cost.detach()), "relu layers", len(rec))
for i, ((m0, pre0), (m2, pre2)) in enumerate(zip(masks[0], masks[2])):
    d = (m0 != m2)
    n = int(d.sum())
    print(f"relu layer {i}: {n} of {d.numel()} mask bits differ between the modes")
    if n:
        idx = d.nonzero()[:5]
        for r_, c_ in idx.tolist():
            print(f"   element ({r_}, {c_}): float64 pre-activation {float(pre0[r_, c_]):.3e} (f32 mode operands) / {float(pre2[r_, c_]):.3e}; "
                  f"largest |pre| of the layer {float(pre0.abs().max()):.2f}")
