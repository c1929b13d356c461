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
         alpha=1.0, split_k=0) -> torch.Tensor:
    """out[M,N] (+)= alpha * a[M,K] @ b[K,N] + bias.  a / b may be transposed views (no copies)."""
    pa, M, K, lda, ta = _mat(a, "a")
    pb, K2, N, ldb, tb = _mat(b, "b")
    if K != K2:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        if accumulate:
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


def to_bf16(x: torch.Tensor, out=None) -> torch.Tensor:
    """bf16 copy (round to nearest even) of a contiguous f32 tensor: the operand copies `gemm_bf16in` reads."""
    _chk(x, "x")
    if not x.is_contiguous() or x.numel() % 8:
        raise ValueError("to_bf16: contiguous tensor with a multiple of 8 elements expected")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    if out.dtype != torch.bfloat16 or out.numel() != x.numel() or not out.is_contiguous() or not out.is_cuda:
        raise ValueError("to_bf16: bad output tensor")
    _lib.call("parrot_to_bf16", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def gemm_bf16in(a_t: torch.Tensor, b: torch.Tensor, out: torch.Tensor, accumulate=False, split_k=0) -> torch.Tensor:
    """out[M,N] (+)= a_t[K,M]^T @ b[K,N] with a_t and b ALREADY bf16 (row-major, strides multiples of 8), f32
    accumulation and f32 `out`: the deferred weight gradients dW = X^T . dPre of a bf16-operand decoder."""
    for x_, n_ in ((a_t, "a_t"), (b, "b")):
        if not (x_.is_cuda and x_.dtype == torch.bfloat16 and x_.dim() == 2 and x_.stride(1) == 1):
            raise ValueError(f"gemm_bf16in: {n_} must be a 2-d bf16 GPU tensor with unit inner stride")
    _chk(out, "out")
    K, M = a_t.shape
    K2, N = b.shape
    if K != K2 or out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm_bf16in: shapes do not match")
    _lib.call("parrot_gemm_bf16in", a_t.data_ptr(), a_t.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
              out.stride(0), M, N, K, int(bool(accumulate)), int(split_k), _stream())
    return out


def gemm16(a: torch.Tensor, b: torch.Tensor, out=None, bias=None, accumulate=False, split_k=0) -> torch.Tensor:
    """out[M,N] (+)= a[M,K] @ b[K,N] (+ bias) with a and b ALREADY bf16 (2-d, row-major or transposed views, no copies),
    f32 accumulation and f32 `out`: parrot_gemm_bf16in_ex.  Strides / K / contiguous extents must be multiples of 8."""
    pa, M, K, lda, ta = _mat16(a, "a")
    pb, K2, N, ldb, tb = _mat16(b, "b")
    if K != K2:
        raise ValueError(f"gemm16: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = (torch.zeros if accumulate else torch.empty)((M, N), device=a.device, dtype=torch.float32)
    _chk(out, "out")
    if out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm16: bad output tensor")
    _lib.call("parrot_gemm_bf16in_ex", pa, lda, ta, pb, ldb, tb, out.data_ptr(), out.stride(0), M, N, K,
              ptr(bias, "bias"), int(bool(accumulate)), int(split_k), _stream())
    return out


def _mat16(t: torch.Tensor, name: str):
    _chk(t, name, torch.bfloat16)
    if t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-D tensor")
    if t.stride(1) == 1 and t.stride(0) >= max(1, t.shape[1]):
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), 0
    if t.stride(0) == 1 and t.stride(1) >= max(1, t.shape[0]):
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(1), 1
    raise ValueError(f"{name}: unsupported strides {t.stride()}")


PRECISION_F32, PRECISION_BF16, PRECISION_BF16X3 = 0, 1, 2


def _default_full_precision() -> int:
    import os
    return PRECISION_F32 if os.environ.get("PARROT_GEMM_PRECISION", "") in ("f32", "0") else PRECISION_BF16X3


_FULL_PRECISION = _default_full_precision()


def full_precision() -> int:
    """The mode f32 models run their batched products in: PRECISION_BF16X3 (f32 operands split into three bf16 terms
    inside the kernel, six bf16 MFMAs per block, f32-grade result; default) or PRECISION_F32 (the f32-input matrix
    instructions; PARROT_GEMM_PRECISION=f32 or set_full_precision)."""
    return _FULL_PRECISION


def set_full_precision(mode: int) -> int:
    """Selects the f32-grade mode (PRECISION_F32 / PRECISION_BF16X3) for this process, library default included;
    returns the previous one."""
    global _FULL_PRECISION
    if mode not in (PRECISION_F32, PRECISION_BF16X3):
        raise ValueError("set_full_precision: PRECISION_F32 or PRECISION_BF16X3")
    prev, _FULL_PRECISION = _FULL_PRECISION, int(mode)
    _lib.call("parrot_set_gemm_precision", int(mode))
    return prev


class gemm_precision:
    """``with gemm_precision(PRECISION_BF16):`` -- operand precision of the batched path of ``gemm`` /
    ``gemm_batched`` inside the block (parrot_set_gemm_precision); the previous mode is restored on exit."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = int(_lib.load().parrot_get_gemm_precision())
        _lib.call("parrot_set_gemm_precision", self.mode)
        return self

    def __exit__(self, *exc):
        _lib.call("parrot_set_gemm_precision", self.prev)
        return False


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
            dW = gemm(x2.t(), dy2)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dW, db, None


def linear(x, W, b=None, act=ACT_NONE):
    return _LinearFn.apply(x, W, b, act)


# ----------------------------------------------------------------------------- SampleRNN sample-level tier (training)
def gemm_gated(a: torch.Tensor, b: torch.Tensor, gate: torch.Tensor, out=None) -> torch.Tensor:
    """out[M,N] = (a[M,K] @ b[K,N]) where gate[M,N] > 0, else 0: the backward of a ReLU layer through its saved
    activation, fused into the product that makes the gradient wrt the layer's output (parrot_gemm_gated)."""
    pa, M, K, lda, ta = _mat(a, "a")
    pb, K2, N, ldb, tb = _mat(b, "b")
    _chk(gate, "gate")
    if K != K2 or gate.shape != (M, N) or gate.stride(1) != 1:
        raise ValueError("gemm_gated: shapes do not match")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    _chk(out, "out")
    if out.shape != (M, N) or out.stride(1) != 1:
        raise ValueError("gemm_gated: bad output tensor")
    _lib.call("parrot_gemm_gated", pa, lda, ta, pb, ldb, tb, out.data_ptr(), out.stride(0), M, N, K,
              gate.data_ptr(), gate.stride(0), _stream())
    return out


def _int32(t: torch.Tensor, name: str, bound=None) -> torch.Tensor:
    """int32 copy of an index tensor; with `bound`, every value must lie in [0, bound): the gather / cross-entropy kernels
    index their tables with these values unchecked, where the reference's Embedding / one-hot (and torch's) raise on a bad
    sample code.  The check costs two reductions and a host read per call (PARROT_SKIP_INDEX_CHECK=1 turns it off)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HipCallError(f"{name}: the parrot_amd product path needs a GPU tensor; there is no CPU fallback")
    t32 = t.to(torch.int32).contiguous()
    if bound is not None and t32.numel() and not _SKIP_INDEX_CHECK:
        lo, hi = int(t32.min()), int(t32.max())
        if lo < 0 or hi >= bound:
            raise IndexError(f"{name}: values in [{lo}, {hi}] outside [0, {bound})")
    return t32


import os as _os
_SKIP_INDEX_CHECK = _os.environ.get("PARROT_SKIP_INDEX_CHECK", "0") not in ("", "0")


class _EmbedSumFn(torch.autograd.Function):
    """y[i] = add[i] + sum_j (E . W1[j*EMB:(j+1)*EMB])[idx[i, j]]: lib.ops.Embedding followed by the bias-free Linear
    SampleLevel.L1_PrevSamples (three_tier.py:486-497) without the [rows, FS*EMB] activation -- the folded table the
    generation loop uses (SampleRnnGenDesc::emb_tbl), forward as a gather-sum, backward as a segmented sum."""

    @staticmethod
    def forward(ctx, E, W1, idx, add):
        Q, EMB = E.shape
        N, J = idx.shape
        D = W1.shape[1]
        if W1.shape[0] != J * EMB:
            raise ValueError("embed_sum: W1 must be [J * EMB, D]")
        E_, W1_ = E.contiguous(), W1.contiguous()
        idx32 = _int32(idx, "idx", bound=Q)
        tbl = torch.empty(J, Q, D, device=E.device, dtype=torch.float32)
        with gemm_precision(PRECISION_F32):  # (a table of 2.6 M entries: nothing to gain from bf16 operands)
            gemm_batched(E_.unsqueeze(0).expand(J, -1, -1), W1_.view(J, EMB, D), tbl)
        y = torch.empty(N, D, device=E.device, dtype=torch.float32)
        if add is not None:
            _chk(add, "add")
            add = add.reshape(N, D)
            if add.stride(1) != 1:
                add = add.contiguous()
        _lib.call("parrot_gather_sum_fwd", tbl.data_ptr(), idx32.data_ptr(), add.data_ptr() if add is not None else None,
                  add.stride(0) if add is not None else 0, y.data_ptr(), y.stride(0), N, J, Q, D, _stream())
        ctx.save_for_backward(E_, W1_, idx32)
        ctx.has_add = add is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        E, W1, idx32 = ctx.saved_tensors
        Q, EMB = E.shape
        N, J = idx32.shape
        D = W1.shape[1]
        dy = dy.reshape(N, D)
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        # rows sorted by (index, row) per position j, and where every bin starts: integer bookkeeping for the segmented sum
        vals, perm = torch.sort(idx32.t().contiguous(), dim=1, stable=True)
        bins = torch.arange(Q + 1, device=E.device, dtype=torch.int32).unsqueeze(0).expand(J, -1).contiguous()
        offs = torch.searchsorted(vals, bins).to(torch.int32).contiguous()
        perm = perm.to(torch.int32).contiguous()
        n_ws = int(_lib.load().parrot_gather_sum_bwd_ws_floats(N, J, Q, D))
        ws = torch.empty(n_ws, device=E.device, dtype=torch.float32)
        dtbl = torch.empty(J, Q, D, device=E.device, dtype=torch.float32)
        _lib.call("parrot_gather_sum_bwd", dy.data_ptr(), dy.stride(0), perm.data_ptr(), offs.data_ptr(), dtbl.data_ptr(),
                  ws.data_ptr(), n_ws, N, J, Q, D, 0, _stream())
        dE = dW1 = None
        with gemm_precision(PRECISION_F32):
            if ctx.needs_input_grad[0]:  # dE = sum_j dtbl[j] . W1_j^T
                t = torch.empty(J, Q, EMB, device=E.device, dtype=torch.float32)
                gemm_batched(dtbl, W1.view(J, EMB, D), t, transB=True)
                dE = t.sum(0)
            if ctx.needs_input_grad[1]:  # dW1_j = E^T . dtbl[j]
                dW1 = torch.empty(J, EMB, D, device=E.device, dtype=torch.float32)
                gemm_batched(E.unsqueeze(0).expand(J, -1, -1), dtbl, dW1, transA=True)
                dW1 = dW1.view(J * EMB, D)
        return dE, dW1, None, (dy if ctx.has_add and ctx.needs_input_grad[3] else None)


def embed_sum(E, W1, idx, add=None):
    """sum_j (E . W1_j)[idx[:, j]] (+ add): see _EmbedSumFn.  E [Q, EMB], W1 [J*EMB, D], idx [N, J] integer, add [N, D]."""
    return _EmbedSumFn.apply(E, W1, idx, add)


class _ReluMlpFn(torch.autograd.Function):
    """logits = relu(relu(x . W2 + b2) . W3 + b3) . W4 + b4 (three_tier.py:499-515) with bias + ReLU in the products'
    epilogues and, backward, the ReLU masks in the epilogues of the dx products (parrot_gemm_gated)."""

    @staticmethod
    def forward(ctx, x, W2, b2, W3, b3, W4, b4):
        x = x.reshape(-1, x.shape[-1])
        if x.stride(1) != 1:
            x = x.contiguous()
        y2 = gemm(x, W2, bias=b2, act=ACT_RELU)
        y3 = gemm(y2, W3, bias=b3, act=ACT_RELU)
        out = gemm(y3, W4, bias=b4)
        ctx.save_for_backward(x, y2, y3, W2, W3, W4)
        return out

    @staticmethod
    def backward(ctx, dl):
        x, y2, y3, W2, W3, W4 = ctx.saved_tensors
        if not dl.is_contiguous():
            dl = dl.contiguous()
        need = ctx.needs_input_grad
        d3 = gemm_gated(dl, W4.t(), y3)
        dW4 = gemm(y3.t(), dl) if need[5] else None
        db4 = colsum(dl) if need[6] else None
        d2 = gemm_gated(d3, W3.t(), y2)
        dW3 = gemm(y2.t(), d3) if need[3] else None
        db3 = colsum(d3) if need[4] else None
        dW2 = gemm(x.t(), d2) if need[1] else None
        db2 = colsum(d2) if need[2] else None
        dx = gemm(d2, W2.t()) if need[0] else None
        return dx, dW2, db2, dW3, db3, dW4, db4


def relu_mlp(x, W2, b2, W3, b3, W4, b4):
    return _ReluMlpFn.apply(x, W2, b2, W3, b3, W4, b4)


class _SoftmaxCeFn(torch.autograd.Function):
    """ce[i] = logsumexp(logits[i]) - logits[i, target[i]] (three_tier.py:565-584), one HIP pass each way."""

    @staticmethod
    def forward(ctx, logits, target):
        _chk(logits, "logits")
        x = logits.reshape(-1, logits.shape[-1])
        if x.stride(1) != 1:
            x = x.contiguous()
        rows, Q = x.shape
        t32 = _int32(target.reshape(-1), "target", bound=Q)
        if t32.numel() != rows:
            raise ValueError("softmax_ce: one target per row expected")
        lse = torch.empty(rows, device=x.device, dtype=torch.float32)
        ce = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.call("parrot_softmax_ce_fwd", x.data_ptr(), x.stride(0), t32.data_ptr(), rows, Q, lse.data_ptr(),
                  ce.data_ptr(), _stream())
        ctx.save_for_backward(x, t32, lse)
        ctx.lshape = logits.shape
        return ce

    @staticmethod
    def backward(ctx, g):
        x, t32, lse = ctx.saved_tensors
        rows, Q = x.shape
        g = g.reshape(-1).to(torch.float32).contiguous()
        dx = torch.empty(rows, Q, device=x.device, dtype=torch.float32)
        _lib.call("parrot_softmax_ce_bwd", x.data_ptr(), x.stride(0), t32.data_ptr(), lse.data_ptr(), g.data_ptr(), rows, Q,
                  dx.data_ptr(), dx.stride(0), _stream())
        return dx.view(ctx.lshape), None


def softmax_ce(logits, target):
    """Per-row softmax cross-entropy with integer targets, natural log: [rows]."""
    return _SoftmaxCeFn.apply(logits, target)


class _WeightNormFn(torch.autograd.Function):
    """W_eff = W * (g / ||W||_2 per output column) (sampleRNN/lib/ops.py:101-110): one HIP pass each way."""

    @staticmethod
    def forward(ctx, W, g):
        _chk(W, "W"); _chk(g, "g")
        if W.dim() != 2 or g.dim() != 1 or g.shape[0] != W.shape[1]:
            raise ValueError("weightnorm_fold: W [K, N], g [N] expected")
        Wc = W if W.stride(1) == 1 else W.contiguous()
        gc = g.contiguous()
        K, N = Wc.shape
        out = torch.empty(K, N, device=W.device, dtype=torch.float32)
        norm = torch.empty(N, device=W.device, dtype=torch.float32)
        ws = torch.empty(int(_lib.load().samplernn_weightnorm_ws_floats(N)), device=W.device, dtype=torch.float32)
        _lib.call("samplernn_weightnorm_fold", Wc.data_ptr(), Wc.stride(0), gc.data_ptr(), out.data_ptr(), N, norm.data_ptr(),
                  ws.data_ptr(), K, N, _stream())
        ctx.save_for_backward(Wc, gc, norm)
        return out

    @staticmethod
    def backward(ctx, d):
        Wc, gc, norm = ctx.saved_tensors
        K, N = Wc.shape
        d = d.to(torch.float32)
        if d.stride(1) != 1:
            d = d.contiguous()
        dW = torch.empty(K, N, device=Wc.device, dtype=torch.float32)
        dg = torch.empty(N, device=Wc.device, dtype=torch.float32)
        ws = torch.empty(int(_lib.load().samplernn_weightnorm_ws_floats(N)), device=Wc.device, dtype=torch.float32)
        _lib.call("samplernn_weightnorm_fold_bwd", Wc.data_ptr(), Wc.stride(0), gc.data_ptr(), norm.data_ptr(), d.data_ptr(),
                  d.stride(0), dW.data_ptr(), N, dg.data_ptr(), ws.data_ptr(), K, N, 0, 0, _stream())
        return dW, dg


def weightnorm_fold(W, g):
    """Effective weight of a weight-normalised Linear: W * (g / ||W||_2 per column), differentiable in W and g."""
    return _WeightNormFn.apply(W, g)


# ----------------------------------------------------------------------------- GRU step / scan
def gru_step_fwd(h, inputs, gate_inputs, Wg, Wc, mask=None):
    """One GatedRecurrent step; returns (h_new, saved) with saved = (z, r, rh, c)."""
    B, H = h.shape
    z, r, rh, c, out = (torch.empty((B, H), device=h.device, dtype=torch.float32) for _ in range(5))
    _lib.call("parrot_gru_step_fwd", ptr(h, "h"), ptr(inputs, "inputs"), ptr(gate_inputs, "gate_inputs"),
              ptr(mask, "mask"), ptr(Wg, "Wg"), ptr(Wc, "Wc"), ptr(out), ptr(z), ptr(r), ptr(rh), ptr(c),
              B, H, _stream())
    return out, (z, r, rh, c)


def gru_step_bwd(dh_out, h, Wg, Wc, saved, mask=None):
    """Returns (dh, d_inputs, d_gate_inputs)."""
    z, r, rh, c = saved
    B, H = h.shape
    dh = torch.empty((B, H), device=h.device, dtype=torch.float32)
    dC = torch.empty((B, H), device=h.device, dtype=torch.float32)
    dG = torch.empty((B, 2 * H), device=h.device, dtype=torch.float32)
    _lib.call("parrot_gru_step_bwd", ptr(dh_out, "dh_out"), ptr(h, "h"), ptr(mask, "mask"), ptr(Wg), ptr(Wc),
              ptr(z), ptr(r), ptr(c), ptr(dh), ptr(dC), ptr(dG), B, H, _stream())
    return dh, dC, dG


class _GruStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, gate_inputs, h, Wc, Wg, mask):
        inputs, gate_inputs, h = inputs.contiguous(), gate_inputs.contiguous(), h.contiguous()
        out, saved = gru_step_fwd(h, inputs, gate_inputs, Wg, Wc, mask)
        ctx.save_for_backward(h, Wc, Wg, mask, *saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, Wc, Wg, mask, z, r, rh, c = ctx.saved_tensors
        dh, dC, dG = gru_step_bwd(dout.contiguous(), h, Wg, Wc, (z, r, rh, c), mask)
        dWc = gemm(rh.t(), dC) if ctx.needs_input_grad[3] else None
        dWg = gemm(h.t(), dG) if ctx.needs_input_grad[4] else None
        return dC, dG, dh, dWc, dWg, None


def gru_step(inputs, gate_inputs, h, Wc, Wg, mask=None):
    """Differentiable single GatedRecurrent step (Wc = state_to_state, Wg = state_to_gates)."""
    return _GruStepFn.apply(inputs, gate_inputs, h, Wc, Wg, mask)


class GruSeqRunner:
    """Owns the workspaces + plan of a GRU scan with up to 4 independent chains (include/parrot_hip.h,
    ParrotGruSeqDesc).  Buffers are allocated once, so the plan's hipGraph can be replayed."""

    def __init__(self, T, B, H, nchain, reverse, device, use_graph=False, with_backward=True):
        import ctypes as C
        self.T, self.B, self.H, self.nchain = T, B, H, nchain
        self.reverse = list(reverse)
        f = dict(device=device, dtype=torch.float32)
        self.h = [torch.zeros((T + 1, B, H), **f) for _ in range(nchain)]
        self.z, self.r, self.rh, self.c = ([torch.empty((T, B, H), **f) for _ in range(nchain)] for _ in range(4))
        self.inputs = [torch.zeros((T, B, H), **f) for _ in range(nchain)]
        self.gate_inputs = [torch.zeros((T, B, 2 * H), **f) for _ in range(nchain)]
        self.mask = None
        self.with_backward = with_backward
        if with_backward:
            self.dh = [torch.zeros((T + 1, B, H), **f) for _ in range(nchain)]
            self.dG = [torch.empty((T, B, 2 * H), **f) for _ in range(nchain)]
            self.dC = [torch.empty((T, B, H), **f) for _ in range(nchain)]
        self.use_graph = use_graph
        self._plan = None
        self._weights = None

    def bind(self, Wg, Wc, mask=None):
        """(Re)creates the plan for these weight tensors (list per chain) / mask [T,B]."""
        import ctypes as C
        key = tuple(w.data_ptr() for w in list(Wg) + list(Wc)) + ((mask.data_ptr(),) if mask is not None else ())
        if self._plan is not None and key == self._weights:
            return
        self.close()
        d = _lib.GruSeqDesc()
        d.T, d.B, d.H, d.nchain, d.use_graph = self.T, self.B, self.H, self.nchain, int(self.use_graph)
        for i in range(self.nchain):
            d.reverse[i] = int(self.reverse[i])
            d.Wg[i], d.Wc[i] = ptr(Wg[i], "Wg"), ptr(Wc[i], "Wc")
            d.inputs[i], d.gate_inputs[i] = self.inputs[i].data_ptr(), self.gate_inputs[i].data_ptr()
            d.h[i] = self.h[i].data_ptr()
            d.z[i], d.r[i], d.rh[i], d.c[i] = (x[i].data_ptr() for x in (self.z, self.r, self.rh, self.c))
            if self.with_backward:
                d.dh[i], d.dG[i], d.dC[i] = self.dh[i].data_ptr(), self.dG[i].data_ptr(), self.dC[i].data_ptr()
        d.mask = ptr(mask, "mask") if mask is not None else None
        self.mask = mask
        plan = C.c_void_p()
        _lib.call("parrot_gru_seq_create", C.byref(d), C.byref(plan))
        self._plan, self._weights = plan, key
        self._keep = (list(Wg), list(Wc), mask)

    def forward(self):
        _lib.call("parrot_gru_seq_fwd", self._plan, _stream())

    def backward(self):
        _lib.call("parrot_gru_seq_bwd", self._plan, _stream())

    def close(self):
        if self._plan is not None:
            _lib.load().parrot_gru_seq_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _GruSeqFn(torch.autograd.Function):
    """GatedRecurrent.apply over a sequence: inputs [T,B,H], gate_inputs [T,B,2H], h0 [B,H]."""

    @staticmethod
    def forward(ctx, inputs, gate_inputs, h0, Wc, Wg, mask, reverse):
        T, B, H = inputs.shape
        run = GruSeqRunner(T, B, H, 1, [reverse], inputs.device, use_graph=False)
        run.inputs[0].copy_(inputs)
        run.gate_inputs[0].copy_(gate_inputs)
        run.h[0][0].copy_(h0)
        run.bind([Wg], [Wc], mask.contiguous() if mask is not None else None)
        run.forward()
        ctx.run = run
        ctx.reverse = reverse
        ctx.save_for_backward(Wc, Wg)
        hs = run.h[0][1:]
        return hs.flip(0) if reverse else hs.clone()

    @staticmethod
    def backward(ctx, dhs):
        run = ctx.run
        Wc, Wg = ctx.saved_tensors
        T, B, H = run.T, run.B, run.H
        run.dh[0].zero_()
        run.dh[0][1:].copy_(dhs.flip(0) if ctx.reverse else dhs)
        run.backward()
        dC, dG = run.dC[0], run.dG[0]
        # saved activations are time-indexed; the state that fed time t is slot s (s = step index)
        hprev = run.h[0][:T]
        if ctx.reverse:
            hprev = hprev.flip(0)  # step s handled time T-1-s
        dWc = gemm(run.rh[0].reshape(T * B, H).t(), dC.reshape(T * B, H)) if ctx.needs_input_grad[3] else None
        dWg = gemm(hprev.reshape(T * B, H).t(), dG.reshape(T * B, 2 * H)) if ctx.needs_input_grad[4] else None
        return dC, dG, run.dh[0][0].clone(), dWc, dWg, None, None


def gru_seq(inputs, gate_inputs, h0, Wc, Wg, mask=None, reverse=False):
    """Differentiable GRU scan; returns states [T,B,H] indexed by time."""
    return _GruSeqFn.apply(inputs.contiguous(), gate_inputs.contiguous(), h0.contiguous(), Wc, Wg, mask, reverse)


class _LstmSeqFn(torch.autograd.Function):
    """LSTM scan (ops.py:461-610): pre_in [T,B,4H] (= x.U + b), s0, c0 [B,H], W [H,4H] -> (s [T,B,H], c [T,B,H])."""

    @staticmethod
    def forward(ctx, pre_in, s0, c0, W):
        import ctypes as C
        T, B, H4 = pre_in.shape
        H = H4 // 4
        f = dict(device=pre_in.device, dtype=torch.float32)
        ws = dict(s=torch.empty(T + 1, B, H, **f), c=torch.empty(T + 1, B, H, **f), gates=torch.empty(T, B, 4 * H, **f),
                  dS=torch.zeros(T + 1, B, H, **f), dc=torch.zeros(B, H, **f), dP=torch.empty(T, B, 4 * H, **f),
                  pre_in=pre_in.contiguous(), W=W.contiguous())
        ws['s'][0].copy_(s0)
        ws['c'][0].copy_(c0)
        d = _lib.LstmSeqDesc()
        d.T, d.B, d.H, d.use_graph = T, B, H, 0
        for k in ('W', 'pre_in', 's', 'c', 'gates', 'dS', 'dc', 'dP'):
            setattr(d, k, ws[k].data_ptr())
        plan = C.c_void_p()
        _lib.call('parrot_lstm_seq_create', C.byref(d), C.byref(plan))
        _lib.call('parrot_lstm_seq_fwd', plan, _stream())
        ctx.ws, ctx.plan, ctx.desc = ws, plan, d
        return ws['s'][1:].clone(), ws['c'][1:].clone()

    @staticmethod
    def backward(ctx, ds, dc_seq):
        ws, T = ctx.ws, ctx.desc.T
        B, H = ctx.desc.B, ctx.desc.H
        ws['dS'].zero_()
        ws['dS'][1:].copy_(ds)
        ws['dc'].zero_()
        if dc_seq is not None and bool((dc_seq[:-1] != 0).any()):
            raise NotImplementedError("gradients into intermediate cell states are not supported")
        if dc_seq is not None:
            ws['dc'].copy_(dc_seq[-1])  # only the final cell state may carry a gradient (TBPTT carry)
        _lib.call('parrot_lstm_seq_bwd', ctx.plan, _stream())
        dP = ws['dP']
        dW = gemm(ws['s'][:T].reshape(T * B, H).t(), dP.reshape(T * B, 4 * H)) if ctx.needs_input_grad[3] else None
        out = (dP, ws['dS'][0].clone(), ws['dc'].clone(), dW)
        _lib.load().parrot_lstm_seq_destroy(ctx.plan)
        return out


def lstm_seq(pre_in, s0, c0, W):
    """Differentiable LSTM scan; returns (s [T,B,H], c [T,B,H])."""
    return _LstmSeqFn.apply(pre_in.contiguous(), s0.contiguous(), c0.contiguous(), W)


# ----------------------------------------------------------------------------- attention step
def gmm_attention_fwd(h1, WattT, batt, kappa_prev, ctx, att_type=0, eps=1e-5, alignment=1.0,
                      sharpening=1.0, timing=1.0):
    B, H = h1.shape
    A = kappa_prev.shape[1]
    _, U, E = ctx.shape
    f = dict(device=h1.device, dtype=torch.float32)
    a, b, k = (torch.empty((B, A), **f) for _ in range(3))
    phi = torch.empty((B, U), **f)
    w = torch.empty((B, E), **f)
    _lib.call("parrot_gmm_attention_fwd", ptr(h1, "h1"), ptr(WattT, "WattT"), ptr(batt, "batt"),
              ptr(kappa_prev, "kappa_prev"), ptr(ctx, "ctx"), ptr(a), ptr(b), ptr(k), ptr(phi), ptr(w),
              B, H, A, U, E, int(att_type), float(eps), float(alignment), float(sharpening), float(timing),
              _stream())
    return a, b, k, phi, w


def gmm_attention_bwd(dw, ctx, a, b, kappa, kappa_prev, WattT, dkappa, dh1, att_type=0, eps=1e-5):
    """dkappa [B,A] is updated in place (carry); dh1 [B,H] is accumulated; returns dp [B,3A]."""
    B, H = dh1.shape
    A = kappa.shape[1]
    _, U, E = ctx.shape
    dp = torch.empty((B, 3 * A), device=dw.device, dtype=torch.float32)
    _lib.call("parrot_gmm_attention_bwd", ptr(dw, "dw"), ptr(ctx, "ctx"), ptr(a), ptr(b), ptr(kappa),
              ptr(kappa_prev), ptr(WattT), ptr(dkappa), ptr(dp), ptr(dh1), B, H, A, U, E, int(att_type),
              float(eps), _stream())
    return dp


# ----------------------------------------------------------------------------- optimiser
def sumsq(x, out=None):
    if out is None:
        out = torch.empty((1,), device=x.device, dtype=torch.float32)
    _lib.call("parrot_sumsq", ptr(x, "x"), x.numel(), ptr(out), _stream())
    return out


def adam_clip_step(param, grad, m, v, gnorm_sq, step, lr=1e-4, clip=9.0, grad_scale=1.0,
                   beta1=0.9, beta2=0.999, eps=1e-8):
    _lib.call("parrot_adam_clip_step", ptr(param, "param"), ptr(grad, "grad"), ptr(m), ptr(v), param.numel(),
              ptr(gnorm_sq) if gnorm_sq is not None else None, float(grad_scale), float(clip), float(lr),
              float(beta1), float(beta2), float(eps), int(step), _stream())


# ----------------------------------------------------------------------------- quantisers
def batch_quantize(x, q_levels=256, q_type="mu-law"):
    """quantize.__batch_quantize on the GPU: x [rows, n] float32 -> int16 (mu-law) / int32 (linear)."""
    _chk(x, "x")
    if x.dim() != 2:
        raise ValueError("batch_quantize expects a 2-D tensor")
    x = x.contiguous()
    rows, n = x.shape
    if q_type == "mu-law":
        mode, out = 0, torch.empty((rows, n), device=x.device, dtype=torch.int16)
    elif q_type == "linear":
        mode, out = 1, torch.empty((rows, n), device=x.device, dtype=torch.int32)
    else:
        raise NotImplementedError(q_type)  # quantize.py:38-42: a-law raises NotImplementedError
    ws = torch.empty((2 * rows,), device=x.device, dtype=torch.float64)
    _lib.call("parrot_batch_quantize", x.data_ptr(), rows, n, x.stride(0), ws.data_ptr(), out.data_ptr(), n,
              mode, int(q_levels), _stream())
    return out


def mu2linear(q):
    """quantize.mu2linear on the GPU: integer class indices -> float32 amplitudes."""
    if not q.is_cuda:
        raise _lib.HipCallError("mu2linear needs a GPU tensor; there is no CPU fallback")
    q32 = q.to(torch.int32).contiguous()
    out = torch.empty(q32.shape, device=q.device, dtype=torch.float32)
    _lib.call("parrot_mu2linear", q32.data_ptr(), q32.numel(), out.data_ptr(), _stream())
    return out


# ----------------------------------------------------------------------------- _simple_norm (model.py:24-34)
def simple_norm_fwd(x, eps=1e-5, out=None, add_into=None):
    """Rows of x [R,N] -> (y, sigma); y may be x itself (in place).  add_into [R,N] gets += y."""
    assert x.dim() == 2
    R, N = x.shape
    y = x if out is None else out
    sigma = torch.empty((R,), device=x.device, dtype=torch.float32)
    _lib.call("parrot_simple_norm_fwd", ptr(x, "x"), N, ptr(y, "y"), N, ptr(sigma), R, N, float(eps),
              ptr(add_into) if add_into is not None else None, N, _stream())
    return y, sigma


def simple_norm_bwd(dy, y, sigma, eps=1e-5, out=None, accumulate=False):
    """dx = J^T dy for y = simple_norm(x); `out` may alias dy or y."""
    R, N = y.shape
    dx = torch.empty_like(y) if out is None else out
    _lib.call("parrot_simple_norm_bwd", ptr(dy, "dy"), N, ptr(y, "y"), N, ptr(sigma), ptr(dx, "dx"), N, R, N,
              float(eps), int(bool(accumulate)), _stream())
    return dx


class _SimpleNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y, sigma = simple_norm_fwd(x2, eps, out=torch.empty_like(x2))
        ctx.save_for_backward(y, sigma)
        ctx.eps, ctx.shape = eps, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        y, sigma = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        return simple_norm_bwd(g2, y, sigma, ctx.eps).view(ctx.shape), None


def simple_norm(x, eps=1e-5):
    """Differentiable `_simple_norm` over the last axis (model.py:24-27)."""
    return _SimpleNormFn.apply(x, eps)
