"""bf16 operand mode (BASELINE configs[3]: "3-layer LSTM h=1536 + attention, bf16"): weights and activations are rounded to
bf16 (nearest even) where they enter a matrix-core product, accumulation / states / gradients / master weights stay f32.

Two kinds of checks:
  * exact-arithmetic checks of the pieces against a torch emulation that rounds the same operands (tight tolerance:
    only the f32 accumulation order differs) -- the bf16 fragment-major weight copy bit for bit, the batched GEMM in all
    four operand layouts;
  * the whole training step (cost, frames, every gradient) against the fp64 oracle with the tolerance a bf16 mantissa
    (8 bits, 2^-9 relative rounding per operand) allows: 5e-3 on the cost, 2e-2 on frames / window state, 5e-2 norm-wise
    per gradient.  The f32 path meets 1e-4 / 1e-3 on the same cases (test_gpu_fullshape.py)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, make_batch, rel_err

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).double()


@pytest.mark.parametrize("rows,cols,mode,lstm_h", [(64, 48, 0, 0), (48, 64, 1, 0), (32, 32, 0, 8), (96, 128, 0, 32),
                                                    (160, 96, 1, 0)])
def test_tile_weights_bf16_layout(dev, rows, cols, mode, lstm_h):
    from parrot_amd import _lib, ops
    g = torch.Generator().manual_seed(rows * 7 + cols)
    W = torch.randn(rows, cols, generator=g).to(dev)
    out = torch.zeros(rows * cols, dtype=torch.bfloat16, device=dev)
    _lib.call('parrot_tile_weights_bf16', W.data_ptr(), rows, cols, cols, out.data_ptr(), mode, lstm_h, ops._stream())
    torch.cuda.synchronize()
    Wb = W.to(torch.bfloat16).cpu()
    got = out.cpu().view(-1, 64, 8)  # [block][lane][u]
    K, N = (rows, cols) if mode == 0 else (cols, rows)
    nch = K // 32
    exp = torch.empty_like(got)
    for ct in range(N // 16):
        for c in range(nch):
            for lane in range(64):
                i, kk = lane & 15, lane >> 4
                ks = torch.arange(8) + 32 * c + 8 * kk
                if mode == 0:
                    col = (i >> 2) * lstm_h + ct * 4 + (i & 3) if lstm_h else ct * 16 + i
                    exp[ct * nch + c, lane] = Wb[ks, col]
                else:
                    exp[ct * nch + c, lane] = Wb[ct * 16 + i, ks]
    assert torch.equal(got.view(torch.int16), exp.view(torch.int16))


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 100), (129, 257, 1000), (640, 384, 63), (96, 3072, 4096)])
def test_gemm_bf16_matches_rounded_operands(dev, ta, tb, M, N, K):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(M + N + K + ta * 2 + tb)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    av, bv = (a.t() if ta else a), (b.t() if tb else b)
    ref = _bf(av.cpu()) @ _bf(bv.cpu()) + bias.cpu().double()
    exact = av.cpu().double() @ bv.cpu().double() + bias.cpu().double()
    with ops.gemm_precision(ops.PRECISION_BF16):
        out = ops.gemm(av, bv, bias=bias)
        acc = ops.gemm(av, bv, out=out.clone(), accumulate=True)
        relu = ops.gemm(av, bv, bias=bias, act=ops.ACT_RELU, split_k=1)
    assert ops._lib.load().parrot_get_gemm_precision() == ops.full_precision()  # restored (the f32-grade default)
    assert_close(out, ref, 2e-5, "bf16 gemm vs rounded-operand product")
    assert_close(acc, 2 * ref - bias.cpu().double(), 2e-5, "accumulate")
    assert_close(relu, ref.clamp_min(0), 2e-5, "relu epilogue")
    assert rel_err(out, exact) > 1e-4, "operands were not rounded: the bf16 path did not run"
    f32 = ops.gemm(av, bv, bias=bias)
    assert_close(f32, exact, 1e-5, "f32 mode after the block")


def test_gemm_batched_bf16(dev):
    from parrot_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(3, 150, 70, generator=g).to(dev)
    b = torch.randn(3, 70, 130, generator=g).to(dev)
    out = torch.empty(3, 150, 130, device=dev)
    with ops.gemm_precision(ops.PRECISION_BF16):
        ops.gemm_batched(a, b, out)
    assert_close(out, _bf(a.cpu()) @ _bf(b.cpu()), 2e-5, "batched")


def _bf16_check(dev, kw, T, B, U, seed, kappa_bias=None):
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=seed, scale_by_fan_in=True)
    if kappa_bias is not None:
        p['/parrot/h1_to_att/fork_kappa.b'].fill_(kappa_bias)
    feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=seed + 1, ragged=True, speaker=cfg['use_speaker'])
    res = {}
    for dt in ('bf16', 'float32'):
        m = Parrot(device=dev, compute_dtype=dt, **kw).allocate()
        m.set_parameter_values(p)
        for rep in range(2):
            m.zero_grad()
            cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev),
                                            None if spk is None else spk.to(dev), 1, B)
            cost.backward()
        res[dt] = (float(cost), [x.detach().cpu().double() for x in av[:3]],
                   {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()})
        m.close()
    for v in p.values():
        v.requires_grad_()
    rc, _, rav, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, spk, 1)
    rc.backward()
    cost, av, grads = res['bf16']
    assert abs(cost - float(rc)) <= 5e-3 * abs(float(rc)), (cost, float(rc))
    assert_close(av[0], rav[0], 2e-2, "predicted frames")
    assert_close(av[1], rav[1], 2e-2, "kappa")
    assert_close(av[2], rav[2], 2e-2, "w")
    worst, differs, n = 0.0, 0, 0
    for name, ref in p.items():
        if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
            continue
        e = rel_err(grads[name], ref.grad)
        assert e <= 5e-2, f"grad {name}: rel err {e:.3e}"
        worst = max(worst, e)
        differs += rel_err(grads[name], res['float32'][2][name]) > 1e-5
        n += 1
    assert n >= 10
    assert differs >= n // 2, "bf16 and f32 gradients coincide: the bf16 path did not run"
    return worst


@pytest.mark.parametrize("cell,L,H", [("gru", 2, 64), ("lstm", 3, 96), ("gru", 1, 32), ("lstm", 1, 64)])
def test_compute_cost_bf16_small(dev, cell, L, H):
    kw = dict(num_layers=L, rnn_h_dim=H, readouts_dim=H, encoder_type='bidirectional', encoder_dim=32, cell_type=cell,
              weak_feedback=True)
    _bf16_check(dev, kw, T=7, B=5, U=11, seed=31 + L)


def test_compute_cost_bf16_cfg4_width(dev):
    """BASELINE configs[3] widths: 3 x LSTM-1536, B = 64 per GPU (the <2,2> / <2,1> bf16 kernels, merged launches)."""
    kw = dict(num_layers=3, rnn_h_dim=1536, readouts_dim=1536, encoder_type='bidirectional', cell_type='lstm')
    _bf16_check(dev, kw, T=4, B=64, U=60, seed=77, kappa_bias=-1.5)


def test_compute_cost_bf16_cfg2_width(dev):
    kw = dict(num_layers=2, rnn_h_dim=1024, readouts_dim=1024, encoder_type='bidirectional', cell_type='gru')
    _bf16_check(dev, kw, T=5, B=64, U=100, seed=78, kappa_bias=-1.5)


def test_bf16_rejects_unsupported_shapes(dev):
    from parrot_amd.model import Parrot
    m = Parrot(device=dev, compute_dtype='bf16', num_layers=1, rnn_h_dim=48, readouts_dim=48,
               encoder_type='bidirectional', encoder_dim=32).allocate()
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(4, 2, 63, generator=g).to(dev)
    with pytest.raises(ValueError):
        m.compute_cost(feat, torch.ones(4, 2, device=dev), torch.zeros(2, 5, dtype=torch.long, device=dev),
                       torch.ones(2, 5, device=dev), None, 1, 2)
    m.close()


@pytest.mark.parametrize("B", [5, 37, 64])
def test_wide_bf16_step_kernel_small(dev, B, monkeypatch):
    """wk_kernel (all rows x 64/128 columns per workgroup, activation stage through LDS) forced onto small LSTM launches:
    partial row blocks (B = 5, 37), several K segments, fwd (fused LSTM epilogue) and bwd (accumulating linear jobs)."""
    monkeypatch.setenv("PARROT_WK", "2")
    kw = dict(num_layers=3, rnn_h_dim=128, readouts_dim=128, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
    _bf16_check(dev, kw, T=5, B=B, U=9, seed=91 + B)


def test_wide_and_tiled_bf16_kernels_agree(dev, monkeypatch):
    """Same bf16 operands, different accumulation order: wk_kernel vs sk_kernel.  A 1e-7 difference in an f32 state can flip
    the bf16 rounding of that element at the next step (4e-3 relative on it), so the two paths agree to bf16 noise, not to
    f32 noise: 5e-3 norm-wise per gradient (the oracle tolerance of the mode is 5e-2), 1e-4 on the cost."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=2, rnn_h_dim=256, readouts_dim=256, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=5, scale_by_fan_in=True)
    feat, fm, lab, lm, spk = make_batch(cfg, 6, 48, 10, seed=6, ragged=True)
    got = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("PARROT_WK", mode)
        m = Parrot(device=dev, compute_dtype='bf16', **kw).allocate()
        m.set_parameter_values(p)
        m.zero_grad()
        cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, 48)
        cost.backward()
        got[mode] = (float(cost), {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()})
        m.close()
    assert abs(got["0"][0] - got["2"][0]) <= 1e-4 * abs(got["0"][0])
    for k, v in got["0"][1].items():
        if float(v.abs().max()) > 1e-12:
            assert rel_err(got["2"][1][k], v) < 5e-3, (k, rel_err(got["2"][1][k], v))


def test_gemm_bf16in_vs_rounded_operands(dev):
    """parrot_to_bf16 + parrot_gemm_bf16in (bf16 operands in memory, 256 x 256 tiles, transposed LDS fragment reads):
    the copy is the round-to-nearest-even of the input bit for bit, and A^T . B equals the float64 product of the rounded
    operands to f32-accumulation accuracy -- ragged M / N (multiples of 8 only), K not a multiple of the K-tile,
    automatic and forced split-K, accumulate."""
    from parrot_amd import ops
    g = torch.Generator().manual_seed(5)
    for K, M, N in ((1000, 264, 520), (4104, 1536, 256), (2048, 8, 776)):
        a = torch.randn(K, M, generator=g).to(dev)
        b = torch.randn(K, N, generator=g).to(dev)
        a16, b16 = ops.to_bf16(a), ops.to_bf16(b)
        assert torch.equal(a16.cpu(), a.cpu().to(torch.bfloat16)) and torch.equal(b16.cpu(), b.cpu().to(torch.bfloat16))
        ref = a16.cpu().double().t() @ b16.cpu().double()
        for split in (0, 1, 3):
            out = torch.full((M, N), 7.0, device=dev)
            ops.gemm_bf16in(a16, b16, out, accumulate=False, split_k=split)
            assert_close(out, ref, 2e-5, f"bf16in K={K} M={M} N={N} split={split}")
            ops.gemm_bf16in(a16, b16, out, accumulate=True, split_k=split)
            assert_close(out, 2 * ref, 2e-5, "accumulate")
        # a view with a leading dimension larger than its width (row slices of a history)
        wide = torch.randn(K, M + 24, generator=g).to(dev)
        w16 = ops.to_bf16(wide)
        out = torch.empty(M, N, device=dev)
        ops.gemm_bf16in(w16[:, 8:8 + M], b16, out)
        assert_close(out, w16.cpu().double()[:, 8:8 + M].t() @ b16.cpu().double(), 2e-5, "strided A")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm16_every_layout_vs_rounded_operands(dev, ta, tb):
    """parrot_gemm_bf16in_ex (round 5: the bf16-in kernel for k-contiguous operands too -- the readout products x . Wr and
    dread . Wr^T of a bf16-operand decoder): every operand layout parrot_gemm takes, f32 bias, accumulate, forced and
    automatic split-K, row slices with a leading dimension larger than their width, M / N not multiples of the 256-wide
    tile, K not a multiple of the K tile -- against the float64 product of the same bf16 values."""
    from parrot_amd import ops
    g = torch.Generator().manual_seed(11 + 2 * ta + tb)
    for M, N, K in ((264, 520, 1000), (1544, 256, 4104), (8, 776, 2048), (640, 1536, 1792)):
        a = torch.randn(M, K, generator=g)
        b = torch.randn(K, N, generator=g)
        bias = torch.randn(N, generator=g).to(dev)
        # memory images: [M, K + 16] or [K, M + 24] for A, [K, N + 8] or [N, K + 32] for B; the operand is a slice
        am = torch.zeros(K, M + 24) if ta else torch.zeros(M, K + 16)
        (am[:, 8:8 + M] if ta else am[:, :K]).copy_(a.t() if ta else a)
        bm = torch.zeros(N, K + 32) if tb else torch.zeros(K, N + 8)
        (bm[:, 16:16 + K] if tb else bm[:, :N]).copy_(b.t() if tb else b)
        a16m, b16m = ops.to_bf16(am.to(dev)), ops.to_bf16(bm.to(dev))
        a16 = a16m[:, 8:8 + M].t() if ta else a16m[:, :K]
        b16 = b16m[:, 16:16 + K].t() if tb else b16m[:, :N]
        assert a16.shape == (M, K) and b16.shape == (K, N)
        ref = a16.cpu().double() @ b16.cpu().double()
        for split in (0, 1, 3):
            out = torch.full((M, N), 7.0, device=dev)
            ops.gemm16(a16, b16, out=out, bias=bias, split_k=split)
            assert_close(out, ref + bias.cpu().double(), 2e-5, f"gemm16 ta={ta} tb={tb} {M}x{N}x{K} split={split}")
            ops.gemm16(a16, b16, out=out, accumulate=True, split_k=split)
            assert_close(out, 2 * ref + bias.cpu().double(), 2e-5, "accumulate")
    with pytest.raises(Exception):  # K must be a multiple of 8 (16-byte vectors along k)
        ops.gemm16(torch.zeros(16, 20, device=dev, dtype=torch.bfloat16), torch.zeros(20, 16, device=dev, dtype=torch.bfloat16))


def test_bf16_readouts_on_copies_match_the_rounding_gemm(dev, monkeypatch):
    """PARROT_BF16_READOUT=1 (default: h . Wr, dread . Wr^T on bf16 copies, parrot_gemm_bf16in_ex) and the f32-operand bf16
    GEMM it replaces round the same values the same way: cost, frames and every gradient agree to f32 summation order."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=3, rnn_h_dim=128, readouts_dim=136, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=9, scale_by_fan_in=True)
    feat, fm, lab, lm, spk = make_batch(cfg, 9, 24, 12, seed=10, ragged=True)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PARROT_BF16_READOUT", mode)
        m = Parrot(device=dev, compute_dtype='bf16', **kw).allocate()
        m.set_parameter_values(p)
        m.zero_grad()
        cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, 24)
        cost.backward()
        got[mode] = (float(cost), av[0].detach().cpu().double().clone(),
                     {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()})
        m.close()
    assert abs(got["0"][0] - got["1"][0]) <= 1e-6 * abs(got["0"][0])
    assert rel_err(got["1"][1], got["0"][1]) < 1e-5
    n = 0
    for k, v in got["0"][2].items():
        if float(v.abs().max()) > 1e-12:
            assert rel_err(got["1"][2][k], v) < 2e-5, (k, rel_err(got["1"][2][k], v))
            n += 1
    assert n >= 10


def test_bf16_weight_grads_match_the_rounding_gemm(dev, monkeypatch):
    """The bf16-copy weight-gradient path (PARROT_BF16_DW=1, default) and the f32-operand bf16 GEMM it replaces round the
    same values the same way: every decoder weight gradient agrees to f32 summation order (1e-5 norm-wise)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=3, rnn_h_dim=128, readouts_dim=128, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=9, scale_by_fan_in=True)
    feat, fm, lab, lm, spk = make_batch(cfg, 9, 24, 12, seed=10, ragged=True)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PARROT_BF16_DW", mode)
        for cell in ("lstm", "gru"):
            kw2 = dict(kw, cell_type=cell)
            cfg2 = R.default_config(**kw2)
            p2 = R.init_params(cfg2, seed=9, scale_by_fan_in=True)
            m = Parrot(device=dev, compute_dtype='bf16', **kw2).allocate()
            m.set_parameter_values(p2)
            m.zero_grad()
            cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, 24)
            cost.backward()
            got[(mode, cell)] = {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()}
            m.close()
    for cell in ("lstm", "gru"):
        n = 0
        for k, v in got[("0", cell)].items():
            if float(v.abs().max()) > 1e-12:
                assert rel_err(got[("1", cell)][k], v) < 1e-5, (cell, k, rel_err(got[("1", cell)][k], v))
                n += 1
        assert n >= 10


def test_bf16_lstm_one_launch_per_tick_agrees_with_schedule_0(dev, monkeypatch):
    """Schedule 7 on the wide bf16 kernel (wk_body's flagged tail; default for bf16 LSTM stacks the wide kernel takes):
    cost, frames, w and every gradient agree with schedule 0 to bf16 noise for partial row blocks, 1-3 layers, graph
    replay; and the plan really ran it."""
    from oracle import parrot_ref as R
    from parrot_amd import _lib
    from parrot_amd.model import Parrot
    monkeypatch.setenv("PARROT_WK", "2")
    for L, B in ((3, 37), (1, 64), (2, 5)):
        kw = dict(num_layers=L, rnn_h_dim=128, readouts_dim=128, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
        cfg = R.default_config(**kw)
        p = R.init_params(cfg, seed=40 + L, scale_by_fan_in=True)
        feat, fm, lab, lm, _ = make_batch(cfg, 7, B, 9, seed=50 + L, ragged=True)
        got = {}
        for sched in ("0", "7"):
            monkeypatch.setenv("PARROT_SCHEDULE", sched)
            m = Parrot(device=dev, compute_dtype='bf16', use_graph=True, **kw).allocate()
            m.set_parameter_values(p)
            for rep in range(2):
                m.zero_grad()
                cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
                cost.backward()
            ws = next(iter(m._train_ws.values()))
            assert int(_lib.load().parrot_decoder_schedule(ws['plan'])) == int(sched)
            got[sched] = (cost.detach().clone(), av[0].detach().clone(), av[2].detach().clone(),
                          {k: v.detach().clone() for k, v in m.get_gradient_dict().items()})
            m.close()
        # Same products of the same rounded operands; the attention runs one block per row instead of column slices, so
        # its sums differ in the last bits and a state can round to the other bf16 neighbour a step later: agreement to
        # bf16 noise (as wk_kernel vs sk_kernel above), far inside the oracle tolerances of the mode.
        assert abs(float(got["0"][0]) - float(got["7"][0])) <= 1e-4 * abs(float(got["0"][0])), (L, B)
        for i, n in ((1, "frames"), (2, "w")):
            assert_close(got["7"][i], got["0"][i].double().cpu(), 2e-3, f"L={L} B={B} {n}")
        for k, v in got["0"][3].items():
            if float(v.abs().max()) > 1e-12:
                assert rel_err(got["7"][3][k], v) < 5e-3, (L, B, k, rel_err(got["7"][3][k], v))


@pytest.mark.parametrize("shape", ["small", "cfg4"])
def test_in_launch_handoffs_never_see_stale_rows(dev, monkeypatch, shape):
    """The fused ticks of schedule 7 hand rows from one workgroup to another INSIDE a launch (forward: the attention's w
    rows; backward: the state-backward's dP rows, read with ordinary L2-cached loads).  A cache line of an earlier
    window surviving in some L2 would be invisible on a replay with the same data -- so one plan (same buffers, captured
    graphs) is replayed on batch A, then on a different batch B, then on A again, and every result must equal what a
    fresh model computes on that batch alone, bit for bit.  "small": shapes at which everything would stay cache-resident;
    "cfg4" (round 5): BASELINE configs[3]'s own widths (3 x LSTM-1536, B = 64, U = 200), where a tick's dP rows
    and weights do not fit any L2 and the launches are the ones the benchmark runs."""
    from oracle import parrot_ref as R
    from parrot_amd import _lib
    from parrot_amd.model import Parrot
    if shape == "small":
        monkeypatch.setenv("PARROT_WK", "2")
        kw = dict(num_layers=3, rnn_h_dim=128, readouts_dim=128, encoder_type='bidirectional', encoder_dim=64, cell_type='lstm')
        T, B, U = 6, 48, 9
    else:
        kw = dict(num_layers=3, rnn_h_dim=1536, readouts_dim=1536, encoder_type='bidirectional', cell_type='lstm')
        T, B, U = 12, 64, 200
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=61, scale_by_fan_in=True)
    batches = [make_batch(cfg, T, B, U, seed=70 + i) for i in range(2)]

    def run(m, batch):
        feat, fm, lab, lm, _ = batch
        m.zero_grad()
        cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
        cost.backward()
        return (cost.detach().clone(), av[0].detach().clone(), {k: v.detach().clone() for k, v in m.get_gradient_dict().items()})

    fresh = []
    for b in batches:
        m = Parrot(device=dev, compute_dtype='bf16', use_graph=True, **kw).allocate()
        m.set_parameter_values(p)
        fresh.append(run(m, b))
        ws = next(iter(m._train_ws.values()))
        assert int(_lib.load().parrot_decoder_schedule(ws['plan'])) == 7
        m.close()
    m = Parrot(device=dev, compute_dtype='bf16', use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    for which in (0, 1, 0, 1, 1, 0):
        cost, frames, grads = run(m, batches[which])
        assert torch.equal(cost, fresh[which][0]) and torch.equal(frames, fresh[which][1]), which
        for k, v in fresh[which][2].items():
            assert torch.equal(grads[k], v), (which, k)
    m.close()
