"""The fused training operators of the SampleRNN sample-level tier (parrot_amd/csrc/trainops.hip, round 5) against
plain torch float64 restatements of the reference's own formulation: Embedding -> reshape -> Linear without bias
(three_tier.py:486-497, sampleRNN/lib/ops.py:252-266), relu(Linear) x 2 -> Linear (:499-515), and
categorical_crossentropy(softmax(logits), target) (:565-584).  The whole-tier parity (cost, ip_cost, every gradient vs the
vectors obtained by EXECUTING the reference's three_tier.compute_cost) is tests/test_gpu_ref_golden.py, which runs on
these operators."""
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _idx(N, J, Q, kind, g):
    if kind == "uniform":
        return torch.randint(0, Q, (N, J), generator=g)
    if kind == "peaked":  # mu-law audio: most samples near the middle code, many codes never used
        x = (torch.randn(N, J, generator=g) * 3 + Q // 2).round().clamp(0, Q - 1).long()
        x[::7] = Q // 2  # one bin far longer than a chunk of the segmented sum
        return x
    return torch.full((N, J), Q - 1, dtype=torch.long)  # "single": one bin holds everything


@pytest.mark.parametrize("N,J,Q,EMB,D,kind", [
    (1000, 10, 256, 16, 64, "uniform"), (777, 10, 256, 8, 32, "peaked"), (130, 3, 5, 4, 1024, "single"),
    (4096, 10, 256, 256, 1024, "peaked"), (515, 10, 256, 32, 2048, "uniform")])
def test_embed_sum_matches_embedding_then_linear(dev, N, J, Q, EMB, D, kind):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(N + D)
    E = torch.randn(Q, EMB, generator=g, dtype=torch.float64)
    W1 = torch.randn(J * EMB, D, generator=g, dtype=torch.float64) / (J * EMB) ** 0.5
    add = torch.randn(N, D, generator=g, dtype=torch.float64)
    idx = _idx(N, J, Q, kind, g)
    dy = torch.randn(N, D, generator=g, dtype=torch.float64)
    ref_in = [t.clone().requires_grad_() for t in (E, W1, add)]
    ref = ref_in[0][idx.reshape(-1)].reshape(N, J * EMB) @ ref_in[1] + ref_in[2]
    ref.backward(dy)
    outs = []
    for rep in range(2):
        hip_in = [t.float().to(dev).requires_grad_() for t in (E, W1, add)]
        y = ops.embed_sum(hip_in[0], hip_in[1], idx.to(dev), hip_in[2])
        y.backward(dy.float().to(dev))
        outs.append([y.detach()] + [t.grad for t in hip_in])
    assert_close(outs[0][0], ref, 2e-6, "embed_sum forward")
    for got, r, n in zip(outs[0][1:], ref_in, ("dE", "dW1", "dadd")):
        assert_close(got, r.grad, 2e-5, n)
    for a, b in zip(*outs):  # fixed summation order: two runs agree bit for bit
        assert torch.equal(a, b)
    # without the additive input
    y0 = ops.embed_sum(E.float().to(dev), W1.float().to(dev), idx.to(dev))
    assert_close(y0, E[idx.reshape(-1)].reshape(N, J * EMB) @ W1, 2e-6, "embed_sum without add")


@pytest.mark.parametrize("N,D,Q", [(300, 64, 256), (1000, 1024, 256), (129, 32, 17)])
def test_relu_mlp_matches_torch(dev, N, D, Q):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(N)
    shapes = [(N, D), (D, D), (D,), (D, D), (D,), (D, Q), (Q,)]
    vals = [torch.randn(*s, generator=g, dtype=torch.float64) / (D ** 0.5 if len(s) == 2 and s[0] == D else 1.0) for s in shapes]
    dl = torch.randn(N, Q, generator=g, dtype=torch.float64)
    r = [v.clone().requires_grad_() for v in vals]
    ref = torch.relu(torch.relu(r[0] @ r[1] + r[2]) @ r[3] + r[4]) @ r[5] + r[6]
    ref.backward(dl)
    h = [v.float().to(dev).requires_grad_() for v in vals]
    out = ops.relu_mlp(*h)
    out.backward(dl.float().to(dev))
    assert_close(out, ref, 1e-5, "logits")
    for got, want, n in zip(h, r, ("dx", "dW2", "db2", "dW3", "db3", "dW4", "db4")):
        assert_close(got.grad, want.grad, 1e-4, n)


def test_gemm_gated_is_a_product_times_the_relu_mask(dev):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(3)
    for M, K, N in ((500, 96, 200), (70, 256, 1024), (4096, 256, 1024)):
        a = torch.randn(M, K, generator=g).to(dev)
        b = torch.randn(N, K, generator=g).to(dev)  # used transposed, like W^T in the backward pass
        gate = torch.randn(M, N, generator=g).to(dev)
        gate[0, :5] = 0.0  # relu(0) = 0 has a zero gradient
        out = ops.gemm_gated(a, b.t(), gate)
        ref = (a.double() @ b.double().t()) * (gate > 0)
        assert_close(out, ref, 1e-5, f"gated {M}x{K}x{N}")
        assert float(out[0, :5].abs().max()) == 0.0


@pytest.mark.parametrize("rows,Q", [(1000, 256), (37, 10), (128000, 256)])
def test_softmax_ce_matches_torch(dev, rows, Q):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, Q, generator=g, dtype=torch.float64) * 4)
    x[0] += 80.0  # large logits: the max shift matters
    t = torch.randint(0, Q, (rows,), generator=g)
    w = torch.rand(rows, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_()
    ref = torch.logsumexp(xr, -1) - xr.gather(1, t[:, None])[:, 0]
    (ref * w).sum().backward()
    xh = x.float().to(dev).requires_grad_()
    ce = ops.softmax_ce(xh, t.to(dev))
    (ce * w.float().to(dev)).sum().backward()
    assert_close(ce, ref, 2e-6, "cross-entropy")
    assert_close(xh.grad, xr.grad, 1e-5, "d logits")


def test_bad_sample_codes_raise(dev):
    """A sample code outside [0, Q) raises (as the reference's Embedding / one-hot do) instead of indexing out of bounds."""
    from parrot_amd import ops
    E = torch.randn(16, 8, device=dev)
    W1 = torch.randn(3 * 8, 32, device=dev)
    idx = torch.randint(0, 16, (40, 3), device=dev)
    ops.embed_sum(E, W1, idx)
    bad = idx.clone()
    bad[7, 1] = 16
    with pytest.raises(IndexError):
        ops.embed_sum(E, W1, bad)
    logits = torch.randn(40, 16, device=dev)
    with pytest.raises(IndexError):
        ops.softmax_ce(logits, torch.full((40,), -1, device=dev))


@pytest.mark.parametrize("K,N", [(1024, 1024), (2560, 1024), (63, 3072), (1024, 10240), (7, 5), (130, 65)])
def test_weightnorm_fold_matches_torch(dev, K, N):
    """samplernn_weightnorm_fold (sampleRNN/lib/ops.py:101-110: W * (g / W.norm(2, axis=0))) and its backward against the
    float64 expression and torch autograd; a strided (column-sliced) W; the SampleRNN Linear built on it."""
    from parrot_amd import ops
    g0 = torch.Generator().manual_seed(K * 31 + N)
    W = torch.randn(K, N, generator=g0, dtype=torch.float64) * 0.3
    gg = torch.rand(N, generator=g0, dtype=torch.float64) + 0.5
    d = torch.randn(K, N, generator=g0, dtype=torch.float64)
    Wr, gr = W.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    ref = Wr * (gr / Wr.norm(2, dim=0)).unsqueeze(0)
    ref.backward(d)
    Wh, gh = W.float().to(dev).requires_grad_(True), gg.float().to(dev).requires_grad_(True)
    out = ops.weightnorm_fold(Wh, gh)
    out.backward(d.float().to(dev))
    assert_close(out.detach().cpu().double(), ref.detach(), 2e-6, "W_eff")
    assert_close(Wh.grad.cpu().double(), Wr.grad, 1e-5, "dW")
    assert_close(gh.grad.cpu().double(), gr.grad, 1e-5, "dg")
    if N >= 64:  # a column slice of a wider matrix: leading dimension > N
        wide = torch.randn(K, N + 40, generator=g0).to(dev)
        sl = wide[:, 8:8 + N]
        o2 = ops.weightnorm_fold(sl, gh.detach())
        r2 = sl.double().cpu() * (gg / sl.double().cpu().norm(2, dim=0)).unsqueeze(0)
        assert_close(o2.cpu().double(), r2, 2e-6, "W_eff of a strided W")


def test_weightnorm_linear_uses_the_fold(dev):
    """lib.ops.effective_weight on device parameters goes through the HIP fold and stays differentiable in W and g."""
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.lib import ops as sops
    lib.delete_all_params()
    lib.set_device(dev)
    try:
        x = torch.randn(6, 48, device=dev)
        y = sops.Linear('wn_test', 48, 80, x, weightnorm=True)
        W, g = lib.param('wn_test.W0'), lib.param('wn_test.g0')
        ref = x.double().cpu() @ (W.detach().double().cpu() * (g.detach().double().cpu() / W.detach().double().cpu().norm(2, dim=0)))
        ref = ref + lib.param('wn_test.b').detach().double().cpu()
        assert_close(y.detach().cpu().double(), ref, 1e-5, "weight-normalised Linear")
        y.sum().backward()
        assert W.grad is not None and g.grad is not None and float(g.grad.abs().max()) > 0
    finally:
        lib.delete_all_params()
