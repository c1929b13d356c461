"""Tensor-level wrappers over the C ABI (include/parrot_hip.h).

torch is used for device memory, streams and autograd bookkeeping only; every arithmetic op here
runs in libparrot_hip.so.  All functions require float32 CUDA(HIP) tensors and raise otherwise.
"""
from __future__ import annotations

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HipCallError(
            f"{name}: the parrot_amd product path needs a GPU tensor (got {type(t).__name__}"
            f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def ptr(t, name="tensor", dtype=torch.float32) -> int:
    if t is None:
        return None
    _chk(t, name, dtype)
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _mat(t: torch.Tensor, name: str):
    """Returns (ptr, rows, cols, ld, trans) for a 2-D tensor that is row-major or a transposed view."""
    _chk(t, name)
    if t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-D tensor")
    if t.stride(1) == 1 and t.stride(0) >= max(1, t.shape[1]):
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), 0
    if t.stride(0) == 1 and t.stride(1) >= max(1, t.shape[0]):
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(1), 1
    if t.shape[0] == 1 or t.shape[1] == 1:
        t = t.contiguous()
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), 0
    raise ValueError(f"{name}: unsupported strides {t.stride()}")


def gemm(a: torch.Tensor, b: torch.Tensor, bias=None, out=None, accumulate=False, act=ACT_NONE,
         alpha=1.0, split_k=1) -> torch.Tensor:
    """out[M,N] (+)= alpha * a[M,K] @ b[K,N] + bias.  a / b may be transposed views (no copies)."""
    pa, M, K, lda, ta = _mat(a, "a")
    pb, K2, N, ldb, tb = _mat(b, "b")
    if K != K2:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        if accumulate or split_k > 1:
            out = torch.zeros((M, N), device=a.device, dtype=torch.float32)
        else:
            out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    _chk(out, "out")
    if out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm: bad output tensor")
    if M == 0 or N == 0:
        return out
    if K == 0:
        if not accumulate:
            out.zero_()
        return out
    _lib.call("parrot_gemm", pa, lda, ta, pb, ldb, tb, out.data_ptr(), out.stride(0), M, N, K,
              ptr(bias, "bias"), float(alpha), int(bool(accumulate)), int(act), 1, 0, 0, 0, int(split_k),
              _stream())
    return out


def gemm_batched(a, b, out, transA=False, transB=False, accumulate=False):
    """a: [nb, M, K] (or [nb, K, M] if transA), b: [nb, K, N] (or [nb, N, K]); contiguous batches."""
    _chk(a, "a"); _chk(b, "b"); _chk(out, "out")
    nb = a.shape[0]
    M, K = (a.shape[2], a.shape[1]) if transA else (a.shape[1], a.shape[2])
    N = b.shape[1] if transB else b.shape[2]
    assert a.stride(2) == 1 and b.stride(2) == 1 and out.stride(2) == 1
    _lib.call("parrot_gemm", a.data_ptr(), a.stride(1), int(transA), b.data_ptr(), b.stride(1), int(transB),
              out.data_ptr(), out.stride(1), M, N, K, None, 1.0, int(bool(accumulate)), 0, nb,
              a.stride(0), b.stride(0), out.stride(0), 1, _stream())
    return out


def colsum(x: torch.Tensor, out=None, accumulate=False) -> torch.Tensor:
    """Column sums of a [M,N] matrix (bias gradients)."""
    _chk(x, "x")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M, N = x2.shape
    if out is None:
        out = torch.empty((N,), device=x.device, dtype=torch.float32)
        accumulate = False
    if M == 0:
        if not accumulate:
            out.zero_()
        return out
    _lib.call("parrot_colsum", x2.data_ptr(), M, N, x2.stride(0), out.data_ptr(), int(bool(accumulate)), _stream())
    return out


class _LinearFn(torch.autograd.Function):
    """y = x . W + b with x [..., in], W [in, out] (Blocks Linear.apply / lib.ops.Linear)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y = gemm(x2, W, bias=b, act=act)
        ctx.save_for_backward(x2, W, y if act != ACT_NONE else None)
        ctx.act = act
        ctx.has_b = b is not None
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], W.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.act == ACT_RELU:
            dy2 = dy2 * (y > 0).to(dy2.dtype)
        elif ctx.act == ACT_TANH:
            dy2 = dy2 * (1 - y * y)
        elif ctx.act == ACT_SIGMOID:
            dy2 = dy2 * y * (1 - y)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy2, W.t()).reshape(ctx.xshape)
        if ctx.needs_input_grad[1]:
            M = x2.shape[0]
            split = 1
            tiles = ((W.shape[0] + 127) // 128) * ((W.shape[1] + 127) // 128)
            while tiles * split < 256 and M // (split * 2) >= 512:
                split *= 2
            dW = gemm(x2.t(), dy2, split_k=split)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dW, db, None


def linear(x, W, b=None, act=ACT_NONE):
    return _LinearFn.apply(x, W, b, act)
