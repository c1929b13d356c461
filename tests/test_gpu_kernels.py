"""Parity of the individual HIP kernels (through the C ABI) against fp64 references."""
import math

import numpy as np
import pytest
import torch

from tests.util import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).float().to(dev)


@pytest.mark.parametrize("M,N,K", [(1, 16, 16), (4, 30, 63), (16, 64, 256), (33, 100, 129), (64, 2048, 1280),
                                   (64, 96, 1028)])
@pytest.mark.parametrize("transB", [False, True])
def test_skinny_gemm(dev, M, N, K, transB):
    from parrot_amd import ops
    a = _rand((M, K), dev, 1, 1.0 / math.sqrt(K))
    b = _rand((N, K) if transB else (K, N), dev, 2)
    bias = _rand((N,), dev, 3)
    bb = b.t() if transB else b
    out = ops.gemm(a, bb, bias=bias)
    ref = a.double().cpu() @ bb.double().cpu() + bias.double().cpu()
    assert_close(out, ref, 2e-6, "skinny gemm")
    # accumulate + activation
    out2 = ops.gemm(a, bb, out=out.clone(), accumulate=True)
    assert_close(out2, ref + ref - bias.double().cpu(), 2e-6, "skinny gemm accumulate")
    out3 = ops.gemm(a, bb, bias=bias, act=ops.ACT_TANH)
    assert_close(out3, torch.tanh(ref), 5e-6, "skinny gemm tanh")


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 63, 77), (300, 260, 515), (1024, 1024, 1024), (65, 7, 3)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_big_gemm_layouts(dev, M, N, K, ta, tb):
    from parrot_amd import ops
    a = _rand((K, M) if ta else (M, K), dev, 4)
    b = _rand((N, K) if tb else (K, N), dev, 5)
    aa = a.t() if ta else a
    bb = b.t() if tb else b
    out = ops.gemm(aa, bb)
    ref = aa.double().cpu() @ bb.double().cpu()
    assert_close(out, ref, 3e-6, "big gemm")


def test_big_gemm_splitk_bias_batched(dev):
    from parrot_amd import ops
    a = _rand((5000, 96), dev, 6)
    b = _rand((5000, 200), dev, 7)
    out = ops.gemm(a.t(), b, split_k=4)
    assert_close(out, a.double().cpu().t() @ b.double().cpu(), 5e-6, "split-k")
    acc = _rand((96, 200), dev, 8)
    out2 = ops.gemm(a.t(), b, out=acc.clone(), accumulate=True, split_k=2)
    assert_close(out2, acc.double().cpu() + a.double().cpu().t() @ b.double().cpu(), 5e-6, "split-k accumulate")
    x = _rand((3, 70, 40), dev, 9)
    y = _rand((3, 40, 50), dev, 10)
    o = torch.empty(3, 70, 50, device=dev)
    ops.gemm_batched(x, y, o)
    assert_close(o, x.double().cpu() @ y.double().cpu(), 3e-6, "batched")
    # strided operands: sub-blocks of larger matrices
    big = _rand((300, 400), dev, 11)
    sub = big[10:210, 20:148]
    w = _rand((128, 64), dev, 12)
    assert_close(ops.gemm(sub, w), sub.double().cpu() @ w.double().cpu(), 3e-6, "strided A")
    assert_close(ops.colsum(big), big.double().cpu().sum(0), 1e-5, "colsum")
    tall = _rand((40000, 70), dev, 13)
    assert_close(ops.colsum(tall), tall.double().cpu().sum(0), 2e-5, "colsum tall")


# ---- split-bf16 products (PARROT_PRECISION_BF16X3, bgs_kernel): f32 operands, six bf16 MFMAs per block ---------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (128, 132, 100), (300, 260, 516), (1024, 1024, 1024), (1000, 520, 72),
                                   (260, 4096, 2052)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_split_gemm_layouts(dev, M, N, K, ta, tb):
    """Every operand layout, ragged edges in all three dimensions: the split kernel holds the tolerance of the f32 kernel."""
    from parrot_amd import ops
    a = _rand((K, M) if ta else (M, K), dev, 4)
    b = _rand((N, K) if tb else (K, N), dev, 5)
    aa = a.t() if ta else a
    bb = b.t() if tb else b
    with ops.gemm_precision(ops.PRECISION_BF16X3):
        out = ops.gemm(aa, bb)
    ref = aa.double().cpu() @ bb.double().cpu()
    assert_close(out, ref, 3e-6, "split gemm")


def test_split_gemm_splitk_bias_accumulate_relu_gate(dev):
    from parrot_amd import ops
    a = _rand((12000, 512), dev, 6)
    b = _rand((12000, 384), dev, 7)
    ref = a.double().cpu().t() @ b.double().cpu()
    acc = _rand((512, 384), dev, 8)
    bias = _rand((384,), dev, 9)
    with ops.gemm_precision(ops.PRECISION_BF16X3):
        o_auto = ops.gemm(a.t(), b)                      # automatic split (slices dealt to XCDs)
        o_8 = ops.gemm(a.t(), b, split_k=8)
        o_3 = ops.gemm(a.t(), b, split_k=3)              # not a multiple of 8: the 2-d grid path
        o_acc = ops.gemm(a.t(), b, out=acc.clone(), accumulate=True, bias=bias, split_k=8)
        o_acc1 = ops.gemm(a.t(), b, out=acc.clone(), accumulate=True, bias=bias, split_k=1)
        x = _rand((700, 300), dev, 10, 1.0 / math.sqrt(300))
        w = _rand((300, 260), dev, 11)
        o_relu = ops.gemm(x, w, bias=_rand((260,), dev, 12), act=ops.ACT_RELU)
        gate = _rand((700, 260), dev, 13)
        o_gate = ops.gemm_gated(x, w, gate)
        o_tanh = ops.gemm(x, w, act=ops.ACT_TANH)        # tanh epilogues stay on the f32 kernel
        assert torch.equal(ops.gemm(a.t(), b, split_k=8), o_8), "split-K result depends on scheduling"
    for o, what in ((o_auto, "auto"), (o_8, "8 slices"), (o_3, "3 slices")):
        assert_close(o, ref, 5e-6, "split gemm split-k " + what)
    full = acc.double().cpu() + ref + bias.double().cpu()
    assert_close(o_acc, full, 5e-6, "split gemm accumulate + bias, split-k")
    assert_close(o_acc1, full, 5e-6, "split gemm accumulate + bias, one slice")
    r2 = x.double().cpu() @ w.double().cpu()
    assert_close(o_relu, torch.relu(r2 + _rand((260,), dev, 12).double().cpu()), 3e-6, "split gemm relu")
    assert_close(o_gate, r2 * (gate.double().cpu() > 0), 3e-6, "split gemm gated")
    assert_close(o_tanh, torch.tanh(r2), 5e-6, "tanh epilogue (f32 kernel)")


def test_split_gemm_error_gate_vs_f32_mfma(dev):
    """The gate of VERDICT r05 item 1c: on the K = T*B weight-gradient product and on same-sign, wide-range operands
    (nothing cancels, every element has a meaningful relative error) the element-wise error of the split kernel against
    float64 is at most 2 x the f32-input MFMA kernel's on the same operands."""
    from parrot_amd import ops
    from tests.util import rel_err_elem
    g = torch.Generator().manual_seed(3)
    x = torch.randn(51200, 512, generator=g).to(dev)
    dy = torch.randn(51200, 256, generator=g).to(dev)
    ref = x.double().t() @ dy.double()
    a2 = ((torch.rand(512, 8192, generator=g) + 0.5) * torch.exp2(torch.randint(-12, 12, (512, 8192), generator=g).float())).to(dev)
    b2 = (torch.rand(8192, 384, generator=g) + 0.5).to(dev)
    ref2 = a2.double() @ b2.double()
    errs = {}
    for mode in (ops.PRECISION_F32, ops.PRECISION_BF16X3):
        with ops.gemm_precision(mode):
            o1 = ops.gemm(x.t(), dy)
            o2 = ops.gemm(a2, b2)
        errs[mode] = (rel_err_elem(o1, ref), float(((o2.double() - ref2).abs() / ref2.abs()).max()), rel_err(o1, ref))
    f, s = errs[ops.PRECISION_F32], errs[ops.PRECISION_BF16X3]
    print("element-wise error vs fp64 (K = 51200 product, same-sign product, norm-wise): f32 MFMA", f, "split bf16", s)
    assert s[0] <= 2 * f[0] and s[1] <= 2 * f[1] and s[2] <= 2 * f[2], (f, s)
    assert s[1] < 2e-6


def test_split_gemm_terms_are_exact(dev):
    """x = x1 + x2 + x3 exactly (three bf16 terms): a product with the identity returns its operand bit for bit, and a
    one-hot gather-sum adds exactly the selected rows -- for normal floats of any magnitude and sign."""
    from parrot_amd import ops
    g = torch.Generator().manual_seed(5)
    v = (torch.randn(384, 256, generator=g) * torch.exp2(torch.randint(-60, 60, (384, 256), generator=g).float())).to(dev)
    eye = torch.eye(256, device=dev)
    with ops.gemm_precision(ops.PRECISION_BF16X3):
        assert torch.equal(ops.gemm(v, eye), v)
        assert torch.equal(ops.gemm(eye, v.t().contiguous()), v.t().contiguous())
        assert torch.equal(ops.gemm(v.t().contiguous().t(), eye), v)


def test_linear_autograd(dev):
    from parrot_amd import ops
    x = _rand((7, 20, 33), dev, 1).requires_grad_()
    W = _rand((33, 50), dev, 2).requires_grad_()
    b = _rand((50,), dev, 3).requires_grad_()
    y = ops.linear(x, W, b)
    (y * y).sum().backward()
    xr, Wr, br = (t.detach().double().cpu().requires_grad_() for t in (x, W, b))
    yr = xr @ Wr + br
    (yr * yr).sum().backward()
    assert_close(y, yr, 3e-6, "linear fwd")
    assert_close(x.grad, xr.grad, 1e-5, "linear dx")
    assert_close(W.grad, Wr.grad, 1e-5, "linear dW")
    assert_close(b.grad, br.grad, 1e-5, "linear db")


@pytest.mark.parametrize("B,H", [(4, 32), (16, 128), (64, 256), (37, 100)])
@pytest.mark.parametrize("use_mask", [False, True])
def test_gru_step_fwd_bwd(dev, B, H, use_mask):
    from oracle import parrot_ref as R
    from parrot_amd import ops
    h = _rand((B, H), dev, 1)
    inp = _rand((B, H), dev, 2)
    gin = _rand((B, 2 * H), dev, 3)
    Wc = _rand((H, H), dev, 4, 1 / math.sqrt(H))
    Wg = _rand((H, 2 * H), dev, 5, 1 / math.sqrt(H))
    mask = (torch.rand(B, generator=torch.Generator().manual_seed(6)) > 0.3).float().to(dev) if use_mask else None
    ts = [t.clone().requires_grad_() for t in (inp, gin, h, Wc, Wg)]
    out = ops.gru_step(*ts, mask)
    gout = _rand((B, H), dev, 7)
    (out * gout).sum().backward()
    rs = [t.detach().double().cpu().requires_grad_() for t in (inp, gin, h, Wc, Wg)]
    ref = R.gru_step(rs[0], rs[1], rs[2], rs[3], rs[4], None if mask is None else mask.double().cpu())
    (ref * gout.double().cpu()).sum().backward()
    assert_close(out, ref, 1e-5, "gru fwd")
    for t, r, n in zip(ts, rs, ("d_inputs", "d_gate_inputs", "dh", "dWc", "dWg")):
        assert_close(t.grad, r.grad, 5e-5, n)


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("use_mask", [False, True])
def test_gru_seq_fwd_bwd(dev, reverse, use_mask):
    from oracle import parrot_ref as R
    from parrot_amd import ops
    T, B, H = 9, 5, 48
    inp = _rand((T, B, H), dev, 1)
    gin = _rand((T, B, 2 * H), dev, 2)
    h0 = _rand((B, H), dev, 3)
    Wc = _rand((H, H), dev, 4, 1 / math.sqrt(H))
    Wg = _rand((H, 2 * H), dev, 5, 1 / math.sqrt(H))
    mask = (torch.rand(T, B, generator=torch.Generator().manual_seed(6)) > 0.3).float().to(dev) if use_mask else None
    ts = [t.clone().requires_grad_() for t in (inp, gin, h0, Wc, Wg)]
    hs = ops.gru_seq(*ts, mask, reverse)
    gout = _rand((T, B, H), dev, 7)
    (hs * gout).sum().backward()
    rs = [t.detach().double().cpu().requires_grad_() for t in (inp, gin, h0, Wc, Wg)]
    mr = None if mask is None else mask.double().cpu()
    if reverse:
        ref = R.gru_scan(rs[0].flip(0), rs[1].flip(0), rs[2], rs[3], rs[4], None if mr is None else mr.flip(0)).flip(0)
    else:
        ref = R.gru_scan(rs[0], rs[1], rs[2], rs[3], rs[4], mr)
    (ref * gout.double().cpu()).sum().backward()
    assert_close(hs, ref, 2e-5, "gru seq fwd")
    for t, r, n in zip(ts, rs, ("d_inputs", "d_gate_inputs", "dh0", "dWc", "dWg")):
        assert_close(t.grad, r.grad, 1e-4, n)


@pytest.mark.parametrize("T,B,H,reverse,use_mask", [(6, 37, 128, False, True), (5, 20, 256, True, False),
                                                   (64, 200, 128, True, True), (7, 16, 16, False, False)])
def test_gru_seq_rowwise_vs_step_launches(dev, monkeypatch, T, B, H, reverse, use_mask):
    """The row-owning scan kernels (rowgru.hip: one launch per direction for the whole sequence, H <= 256) against the
    fp64 oracle and against the per-step launch path (PARROT_GRU_ROWWISE=0) -- the encoder's own shape (64 steps over
    200 rows, H = 128), ragged row blocks, both directions, step masks, the widest and the narrowest layer it takes."""
    from oracle import parrot_ref as R
    from parrot_amd import ops
    inp = _rand((T, B, H), dev, 1)
    gin = _rand((T, B, 2 * H), dev, 2)
    h0 = _rand((B, H), dev, 3)
    Wc = _rand((H, H), dev, 4, 1 / math.sqrt(H))
    Wg = _rand((H, 2 * H), dev, 5, 1 / math.sqrt(H))
    mask = (torch.rand(T, B, generator=torch.Generator().manual_seed(6)) > 0.3).float().to(dev) if use_mask else None
    gout = _rand((T, B, H), dev, 7)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PARROT_GRU_ROWWISE", mode)
        ts = [t.clone().requires_grad_() for t in (inp, gin, h0, Wc, Wg)]
        hs = ops.gru_seq(*ts, mask, reverse)
        (hs * gout).sum().backward()
        got[mode] = [hs.detach().clone()] + [t.grad.clone() for t in ts]
    rs = [t.detach().double().cpu().requires_grad_() for t in (inp, gin, h0, Wc, Wg)]
    mr = None if mask is None else mask.double().cpu()
    if reverse:
        ref = R.gru_scan(rs[0].flip(0), rs[1].flip(0), rs[2], rs[3], rs[4], None if mr is None else mr.flip(0)).flip(0)
    else:
        ref = R.gru_scan(rs[0], rs[1], rs[2], rs[3], rs[4], mr)
    (ref * gout.double().cpu()).sum().backward()
    names = ("h", "d_inputs", "d_gate_inputs", "dh0", "dWc", "dWg")
    for a, r, n in zip(got["1"], [ref] + [x.grad for x in rs], names):
        assert_close(a, r, 2e-5 if n == "h" else 1e-4, "row-wise vs oracle: " + n)
    for a, b, n in zip(got["1"], got["0"], names):
        assert_close(a, b, 2e-5, "row-wise vs step launches: " + n)


@pytest.mark.parametrize("T,B,H", [(5, 3, 16), (7, 37, 64)])
def test_lstm_seq_fwd_bwd(dev, T, B, H):
    """LSTM scan (ops.py:461-553 arithmetic) vs an fp64 torch loop, values and all gradients."""
    from parrot_amd import ops
    pre = _rand((T, B, 4 * H), dev, 1)
    s0, c0 = _rand((B, H), dev, 2), _rand((B, H), dev, 3)
    W = _rand((H, 4 * H), dev, 4, 1 / math.sqrt(H))
    ts = [t.clone().requires_grad_() for t in (pre, s0, c0, W)]
    s, c = ops.lstm_seq(*ts)
    gs, gc = _rand((T, B, H), dev, 5), _rand((B, H), dev, 6)
    ((s * gs).sum() + (c[-1] * gc).sum()).backward()
    rp, rs, rc, rW = (t.detach().double().cpu().requires_grad_() for t in (pre, s0, c0, W))
    hs, cs, hh, cc = [], [], rs, rc
    for t in range(T):
        z = hh @ rW + rp[t]
        i, f, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.sigmoid(z[:, 2 * H:3 * H])
        g = torch.tanh(z[:, 3 * H:])
        cc = cc * f + g * i
        hh = torch.tanh(cc) * o
        hs.append(hh); cs.append(cc)
    rs_all, rc_all = torch.stack(hs), torch.stack(cs)
    ((rs_all * gs.double().cpu()).sum() + (rc_all[-1] * gc.double().cpu()).sum()).backward()
    assert_close(s, rs_all, 2e-5, "s")
    assert_close(c, rc_all, 2e-5, "c")
    for t, r, n in zip(ts, (rp, rs, rc, rW), ("d_pre", "ds0", "dc0", "dW")):
        assert_close(t.grad, r.grad, 1e-4, n)


@pytest.mark.parametrize("att_type", ["graves", "softmax"])
@pytest.mark.parametrize("B,H,A,U,E", [(4, 64, 10, 23, 32), (64, 256, 10, 200, 256), (3, 40, 5, 17, 420)])
def test_attention_fwd_bwd(dev, att_type, B, H, A, U, E):
    from oracle import parrot_ref as R
    from parrot_amd import ops
    cfg = R.default_config(attention_type=att_type, attention_size=A, sharpening_coeff=1.3, timing_coeff=0.8,
                           attention_alignment=0.7)
    h1 = _rand((B, H), dev, 1)
    Watt = _rand((H, 3 * A), dev, 2, 0.5 / math.sqrt(H))
    batt = _rand((3 * A,), dev, 3, 0.1)
    kprev = _rand((B, A), dev, 4).abs() * 3
    ctx = _rand((B, U, E), dev, 5)
    at = 1 if att_type == "softmax" else 0
    WattT = Watt.t().contiguous()
    a, b, k, phi, w = ops.gmm_attention_fwd(h1, WattT, batt, kprev, ctx, at, 1e-5, 0.7, 1.3, 0.8)
    rh1, rW, rb, rk, rctx = (t.double().cpu().requires_grad_() for t in (h1, Watt, batt, kprev, ctx))
    p = rh1 @ rW + rb
    ra, rkk, rphi, rw = R.attention_step(cfg, p[:, :A], p[:, A:2 * A], p[:, 2 * A:], rk, rctx, sampling=True)
    assert_close(a, ra, 1e-5, "a")
    assert_close(k, rkk, 1e-5, "kappa")
    assert_close(phi, rphi, 2e-5, "phi")
    assert_close(w, rw, 2e-5, "w")
    # backward: L = sum(w * gw) + sum(kappa * gk)
    gw = _rand((B, E), dev, 6)
    gk = _rand((B, A), dev, 7)
    ((rw * gw.double().cpu()).sum() + (rkk * gk.double().cpu()).sum()).backward()
    dkappa = gk.clone()
    dh1 = torch.zeros(B, H, device=dev)
    dp = ops.gmm_attention_bwd(gw, ctx, a, b, k, kprev, WattT, dkappa, dh1, at, 1e-5)
    assert_close(dh1, rh1.grad, 1e-4, "dh1")
    assert_close(dkappa, rk.grad, 1e-4, "dkappa_prev")
    assert_close(dp.sum(0), rb.grad, 1e-4, "dp (bias grad)")


def test_quantize_bit_exact(dev):
    from oracle import quantize_ref as Q
    from parrot_amd import ops
    rng = np.random.RandomState(1234)
    for rows, n in [(1, 3), (4, 1000), (32, 16000), (3, 80)]:
        x = rng.randn(rows, n).astype(np.float32)
        if n == 3:
            x = np.array([[-1, 0, 1]], dtype=np.float32)
        xt = torch.from_numpy(x).to(dev)
        mu = ops.batch_quantize(xt, 256, "mu-law").cpu().numpy()
        ref = Q.batch_quantize(x, 256, "mu-law")
        assert mu.dtype == np.int16 and ref.dtype == np.int16
        assert np.array_equal(mu, ref), f"mu-law mismatch: {(mu != ref).sum()} of {mu.size}"
        lin = ops.batch_quantize(xt, 256, "linear").cpu().numpy()
        refl = Q.batch_quantize(x, 256, "linear")
        assert lin.dtype == np.int32 and np.array_equal(lin, refl)
    q = torch.arange(256, dtype=torch.int32, device=dev)
    dec = ops.mu2linear(q).cpu().numpy()
    refd = Q.mu2linear(np.arange(256, dtype=np.int32))
    assert dec.dtype == np.float32
    np.testing.assert_allclose(dec, refd, rtol=2e-6, atol=1e-8)
    with pytest.raises(NotImplementedError):
        ops.batch_quantize(xt, 256, "a-law")


def test_quantize_golden(dev):
    import os
    from parrot_amd import ops
    path = os.path.join(os.path.dirname(__file__), "golden", "quantize_golden.npz")
    g = np.load(path)
    x = torch.from_numpy(g["x"]).to(dev)
    assert np.array_equal(ops.batch_quantize(x, 256, "mu-law").cpu().numpy(), g["mu"])
    assert np.array_equal(ops.batch_quantize(x, 256, "linear").cpu().numpy(), g["lin"])
    np.testing.assert_allclose(ops.mu2linear(torch.from_numpy(g["mu"].astype(np.int32)).to(dev)).cpu().numpy(),
                               g["dec"], rtol=2e-6, atol=1e-8)


def test_adam_clip(dev):
    from oracle import parrot_ref as R
    from parrot_amd import ops
    n = 10007
    p = _rand((n,), dev, 1)
    for clip_big in (False, True):
        g = _rand((n,), dev, 2, 10.0 if clip_big else 0.01)
        m = torch.zeros(n, device=dev)
        v = torch.zeros(n, device=dev)
        pp = p.clone()
        rp, rg = {"x": p.double().cpu().clone()}, {"x": g.double().cpu()}
        rm, rv = {"x": torch.zeros(n, dtype=torch.float64)}, {"x": torch.zeros(n, dtype=torch.float64)}
        for step in (1, 2, 3):
            nrm = ops.sumsq(g)
            ops.adam_clip_step(pp, g, m, v, nrm, step, lr=1e-2, clip=9.0)
            R.clip_adam_step(rp, rg, rm, rv, step, lr=1e-2, clip=9.0)
        assert_close(pp, rp["x"], 1e-5, "adam params")
        assert_close(m, rm["x"], 1e-5, "adam m")


def test_cpu_tensor_fails_loudly():
    from parrot_amd import _lib, ops
    with pytest.raises(_lib.HipCallError):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))


@pytest.mark.parametrize("R,N", [(1, 7), (5, 64), (33, 1000), (70, 3072)])
def test_simple_norm_fwd_bwd(dev, R, N):
    """_simple_norm (model.py:24-27) forward and backward vs the fp64 oracle + autograd."""
    from oracle import parrot_ref as Rf
    from parrot_amd import ops
    g = torch.Generator().manual_seed(R * 1000 + N)
    x = (torch.randn(R, N, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_()
    dy = torch.randn(R, N, generator=g, dtype=torch.float64)
    y = Rf.simple_norm(x)
    y.backward(dy)
    acc = torch.randn(R, N, generator=g).to(dev)
    acc0 = acc.clone()
    xd = x.detach().float().to(dev)
    yd, sig = ops.simple_norm_fwd(xd.clone(), add_into=acc)  # in place + accumulate
    assert_close(yd, y, 2e-5, "norm fwd")
    assert_close(acc - acc0, y, 2e-5, "norm add_into")
    dx = ops.simple_norm_bwd(dy.float().to(dev), yd, sig)
    assert_close(dx, x.grad, 1e-4, "norm bwd")
    xa = xd.clone().requires_grad_()
    ops.simple_norm(xa).backward(dy.float().to(dev))
    assert_close(xa.grad, x.grad, 1e-4, "norm autograd")


@pytest.mark.parametrize("rows,cols,mode,lstm_h", [(32, 48, 0, 0), (48, 32, 1, 0), (32, 64, 0, 16), (2304, 2048, 0, 0)])
def test_tile_weights_layout(dev, rows, cols, mode, lstm_h):
    """parrot_tile_weights: 256-float blocks [column tile][chunk] holding [kk][i][u] (bit-exact copy)."""
    import ctypes as C
    from parrot_amd import _lib, ops
    g = torch.Generator().manual_seed(rows + cols + mode)
    W = torch.randn(rows, cols, generator=g)
    Wd = W.to(dev)
    out = torch.empty_like(Wd)
    _lib.call('parrot_tile_weights', Wd.data_ptr(), rows, cols, cols, out.data_ptr(), mode, lstm_h, ops._stream())
    got = out.cpu().reshape(-1)
    Wn = W.numpy()
    nct, nch = (cols // 16, rows // 16) if mode == 0 else (rows // 16, cols // 16)
    ref = np.empty((nct, nch, 4, 16, 4), dtype=np.float32)
    for ct in range(nct):
        for i in range(16):
            col = ((i >> 2) * lstm_h + ct * 4 + (i & 3)) if lstm_h else ct * 16 + i
            for kk in range(4):
                for u in range(4):
                    if mode == 0:
                        ref[ct, :, kk, i, u] = Wn[np.arange(nch) * 16 + kk * 4 + u, col]
                    else:
                        ref[ct, :, kk, i, u] = Wn[ct * 16 + i, np.arange(nch) * 16 + kk * 4 + u]
    assert np.array_equal(got.numpy(), ref.reshape(-1))
