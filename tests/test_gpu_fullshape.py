"""Oracle parity AT THE SHAPES THE BENCHMARKS RUN: the small-shape tests select
`sk_kernel<1,1>` with row-major weights, the benchmarks run `<2,2>` / `<2,1>` with fragment-major weight copies and
the merged multi-job launches.  Every test here compares the HIP path with the oracle restatement -- cost, frames,
window state AND every parameter gradient -- at the real widths, on windows short enough for the CPU oracle.

  cfg2  configs[1]: L=2 GRU, H=R=1024, B=64, U=200, ragged masks, kappa bias -1.5 (window stays inside the text)
  cfg4  configs[3]: L=3 LSTM, H=R=1536, B=64 per GPU
  cfg3  configs[2]: decode N=16, H=1024, feedback on, 60 steps
  cfg5  configs[4]: SampleRNN generator DIM=1024, B=32, 2000 samples (bench.py's utterance), greedy

Reference lines: model.py:651-824 (training scan + cost), :882-1057 (decode scan), three_tier.py:795-832."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, make_batch, rel_err, rel_err_elem

pytestmark = pytest.mark.gpu


def _full_check(dev, kw, T, B, U, seed, tol_grad=1e-3, kappa_bias=None, use_graph=True):
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=seed, scale_by_fan_in=True)
    if kappa_bias is not None:
        p['/parrot/h1_to_att/fork_kappa.b'].fill_(kappa_bias)
    m = Parrot(device=dev, use_graph=use_graph, **kw).allocate()
    m.set_parameter_values(p)
    feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=seed + 1, ragged=True, speaker=cfg['use_speaker'])
    for v in p.values():
        v.requires_grad_()
    rc, _, rav, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, spk, 1)
    rc.backward()
    worst = {}
    for rep in range(2):  # second pass replays the captured hipGraphs
        m.zero_grad()
        cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev),
                                        None if spk is None else spk.to(dev), 1, B)
        cost.backward()
        assert_close(cost, rc, 1e-4, "cost")
        assert_close(av[0], rav[0], 1e-4, "predicted frames")
        assert_close(av[1], rav[1], 1e-4, "kappa")
        assert_close(av[2], rav[2], 1e-4, "w")
        assert_close(av[4], rav[4], 1e-4, "phi")
        grads = m.get_gradient_dict()
        n_checked = 0
        for name, ref in p.items():
            if ref.grad is None:
                continue
            scale = float(ref.grad.abs().max())
            if scale < 1e-12:
                assert float(grads[name].abs().max()) < 1e-6, name
                continue
            e = rel_err(grads[name], ref.grad)
            assert e <= tol_grad, f"pass {rep}: grad {name}: rel err {e:.3e} > {tol_grad:.0e}"
            worst[name] = max(worst.get(name, 0.0), e)
            n_checked += 1
        assert n_checked >= 10
    m.close()
    return worst


def test_cfg2_width_cost_and_every_gradient(dev):
    """BASELINE configs[1] widths.  T=6 -> wavefront ticks with one and with two active layers, i.e. both the
    `<2,1>` and the `<2,2>` tile shapes, the BWD_RH epilogue and the fragment-major weight copies."""
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024)
    _full_check(dev, kw, T=6, B=64, U=200, seed=11, kappa_bias=-1.5)


def test_cfg2_width_weak_feedback(dev):
    """Same widths with teacher-forcing feedback (the realistic training configuration)."""
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024, weak_feedback=True)
    _full_check(dev, kw, T=5, B=64, U=200, seed=17, kappa_bias=-1.5)


def test_cfg4_width_lstm1536_cost_and_every_gradient(dev):
    """BASELINE configs[3] widths in fp32 (3 x LSTM-1536 + attention, 64 rows per GPU)."""
    kw = dict(num_layers=3, encoder_type='bidirectional', rnn_h_dim=1536, readouts_dim=1536, cell_type='lstm')
    _full_check(dev, kw, T=4, B=64, U=120, seed=23, kappa_bias=-1.0)


def test_cfg2_benchmarked_window_T800_matches_oracle(dev, capsys):
    """BASELINE configs[1] EXACTLY as bench.py runs it -- L=2 GRU, H=R=1024, B=64, T_enc=200, **T_dec=800**, the
    train.py:30-31 initialisation with the kappa bias at -1.5, full masks -- against the fp64 oracle (checkpointed BPTT,
    oracle/parrot_ref.cost_and_grads_checkpointed): cost, predicted frames, kappa, w, phi at 1e-4 (north star), every
    parameter gradient at 1e-3 norm-wise.  kappa is an 800-term running sum of exp(.), h an 800-deep fp32 recurrence:
    this is the test that says the benchmarked window itself is right, not only short ones.  Prints the norm-wise and
    the element-wise relative errors."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024)
    T, B, U = 800, 64, 200
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=1234)  # N(0, 0.01) weights, zero biases: bench.py's model
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=77)
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    grads = m.get_gradient_dict()
    for v in p.values():
        v.requires_grad_()
    rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=100)
    report = [f"cost: hip {float(cost):.8f} oracle {float(rc):.8f} rel {abs(float(cost) - float(rc)) / abs(float(rc)):.2e}"]
    assert float(rav[1][-1].min()) > 100.0, "kappa must have moved through the text for the window to mean anything"
    for i, n in ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        e = assert_close(av[i], rav[i], 1e-4, n)
        report.append(f"{n}: norm-wise {e:.2e}, element-wise {rel_err_elem(av[i], rav[i]):.2e}")
    assert_close(cost, rc, 1e-4, "cost")
    worst, n_checked = ("", 0.0), 0
    for name, ref in p.items():
        if ref.grad is None:
            continue
        scale = float(ref.grad.abs().max())
        if scale < 1e-12:
            assert float(grads[name].abs().max()) < 1e-6, name
            continue
        e = rel_err(grads[name], ref.grad)
        assert e <= 1e-3, f"grad {name}: rel err {e:.3e} (element-wise {rel_err_elem(grads[name], ref.grad):.3e})"
        if e > worst[1]:
            worst = (name, e)
        n_checked += 1
    assert n_checked >= 10
    report.append(f"{n_checked} parameter gradients within 1e-3; worst {worst[0]}: {worst[1]:.2e}")
    with capsys.disabled():
        print("\n[T800 parity] " + "\n[T800 parity] ".join(report))
    m.close()


@pytest.mark.timeout(2400)
def test_reference_literal_3gru_window_T800_matches_oracle(dev, capsys):
    """The reference's OWN decoder depth -- three GatedRecurrent layers, h = 1024 (model.py:312-347), literal batch-axis encoder
    -- at the benchmarked window (B = 64, T_enc = 200, T_dec = 800, fp32), as `bench.py`'s `secondary.ref_literal_3gru` runs it
    (round 5): cost, frames, kappa, w, phi at 1e-4 and every parameter gradient at 1e-3 norm-wise against the fp64 oracle with
    checkpointed BPTT.  Three layers run schedule 5 forward and the three-launch backward tick (bwd8 covers two layers only)."""
    from oracle import parrot_ref as R
    from parrot_amd import _lib
    from parrot_amd.model import Parrot
    kw = dict(num_layers=3, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024)
    T, B, U = 800, 64, 200
    cfg = R.default_config(**kw)
    assert cfg['encoder_literal']
    p = R.init_params(cfg, seed=1234)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=77)
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    assert int(_lib.load().parrot_decoder_schedule(next(iter(m._train_ws.values()))['plan'])) == 5
    assert int(_lib.load().parrot_decoder_backward_tick(next(iter(m._train_ws.values()))['plan'])) == 8  # bwd8 at L = 3 (round 6)
    grads = {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()}
    av = [x.detach().cpu().double() for x in av]
    cost = float(cost)
    m.close()
    for v in p.values():
        v.requires_grad_()
    rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=100)
    report = [f"cost: hip {cost:.8f} oracle {float(rc):.8f} rel {abs(cost - float(rc)) / abs(float(rc)):.2e}"]
    assert abs(cost - float(rc)) <= 1e-4 * abs(float(rc))
    assert float(rav[1][-1].min()) > 100.0, "kappa must have moved through the text for the window to mean anything"
    for i, n in ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        e = assert_close(av[i], rav[i], 1e-4, n)
        report.append(f"{n}: norm-wise {e:.2e}, element-wise {rel_err_elem(av[i], rav[i]):.2e}")
    worst, n_checked = ("", 0.0), 0
    for name, ref in p.items():
        if ref.grad is None:
            continue
        if float(ref.grad.abs().max()) < 1e-12:
            assert float(grads[name].abs().max()) < 1e-6, name
            continue
        e = rel_err(grads[name], ref.grad)
        assert e <= 1e-3, f"grad {name}: rel err {e:.3e}"
        if e > worst[1]:
            worst = (name, e)
        n_checked += 1
    assert n_checked >= 15
    report.append(f"{n_checked} parameter gradients within 1e-3; worst {worst[0]}: {worst[1]:.2e}")
    with capsys.disabled():
        print("\n[ref-literal 3xGRU T800 parity] " + "\n[ref-literal 3xGRU T800 parity] ".join(report))


# SURVEY 8d variants of the benchmarked windows: (decoder kwargs, init kwargs, kappa bias, B, U, ragged)
VARIANTS = {
    # configs[1] with the N(0, 1/fan_in) parameter set (gates and attention leave their linear regime), ragged lengths
    # T ~ U[600, 800] / U ~ U[120, 200] and teacher-forcing feedback
    "cfg2v": (dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024, weak_feedback=True),
              dict(scale_by_fan_in=True), -1.8, 64, 200, True),
    # configs[3] per GPU: 3 x LSTM-1536, bf16 operands, bench.py's initialisation, full masks
    "cfg4": (dict(num_layers=3, encoder_type='bidirectional', rnn_h_dim=1536, readouts_dim=1536, cell_type='lstm'),
             dict(), -1.5, 64, 200, False),
}


def variant_batch(cfg, T, B, U, ragged, seed):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(T + 1, B, cfg['output_dim'], generator=g, dtype=torch.float64)
    fm = torch.ones(T + 1, B, dtype=torch.float64)
    lab = torch.randint(0, cfg['num_characters'], (B, U), generator=g)
    lm = torch.ones(B, U, dtype=torch.float64)
    if ragged:  # T ~ U[0.75 T, T], U ~ U[0.6 U, U]
        for b in range(B):
            fm[int(torch.randint(3 * T // 4, T + 1, (1,), generator=g)) + 1:, b] = 0
            lm[b, int(torch.randint(3 * U // 5, U + 1, (1,), generator=g)):] = 0
        fm[:, 0] = 1  # one utterance of full length, so the window really is T frames long
    return feat, fm, lab, lm


def _window_check(dev, which, T, compute_dtype, tol_out, tol_cost, tol_grad, capsys, tag):
    """One training window of VARIANTS[which] on the HIP path vs the fp64 oracle (checkpointed BPTT), and -- as the
    yardstick for what ANY float32 evaluation of this map can meet -- the oracle in float32 vs itself in float64."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw, init_kw, kb, B, U, ragged = VARIANTS[which]
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=1234, **init_kw)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(kb)
    feat, fm, lab, lm = variant_batch(cfg, T, B, U, ragged, seed=77)
    m = Parrot(device=dev, use_graph=True, compute_dtype=compute_dtype, **kw).allocate()
    m.set_parameter_values(p)
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    grads = {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()}
    av = [x.detach().cpu().double() for x in av]
    cost = float(cost)
    m.close()
    p32 = {k: v.float().requires_grad_() for k, v in p.items()}
    for v in p.values():
        v.requires_grad_()
    rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=100)
    c32, av32 = R.cost_and_grads_checkpointed(p32, cfg, feat.float(), fm.float(), lab, lm.float(), None, chunk=100)
    rep = [f"{which} T_dec={T} B={B} U={U} ragged={ragged} operands={compute_dtype}",
           f"cost: hip {cost:.8f} oracle {float(rc):.8f} rel {abs(cost - float(rc)) / abs(float(rc)):.2e} "
           f"(oracle-f32 {abs(float(c32) - float(rc)) / abs(float(rc)):.2e})"]
    assert abs(cost - float(rc)) <= tol_cost * abs(float(rc))
    for i, n in ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        e, e32 = rel_err(av[i], rav[i]), rel_err(av32[i], rav[i])
        rep.append(f"{n}: norm-wise {e:.2e} (oracle-f32 {e32:.2e}), element-wise {rel_err_elem(av[i], rav[i]):.2e} "
                   f"(oracle-f32 {rel_err_elem(av32[i], rav[i]):.2e})")
    worst, worst32, n_checked = ("", 0.0), 0.0, 0
    for name, ref in p.items():
        if ref.grad is None:
            continue
        if float(ref.grad.abs().max()) < 1e-12:
            assert float(grads[name].abs().max()) < 1e-6, name
            continue
        e = rel_err(grads[name], ref.grad)
        worst32 = max(worst32, rel_err(p32[name].grad, ref.grad))
        if e > worst[1]:
            worst = (name, e)
        n_checked += 1
    rep.append(f"{n_checked} parameter gradients; worst {worst[0]}: {worst[1]:.2e} norm-wise "
               f"(oracle-f32 worst {worst32:.2e})")
    with capsys.disabled():
        print(f"\n[{tag}] " + f"\n[{tag}] ".join(rep))
    # The bar: the stated tolerance -- or, where the oracle evaluated in float32 itself misses it (an 800-deep recurrence
    # with saturating gates amplifies rounding: a property of the map, not of the kernels), three times the oracle's
    # own float32 error on that tensor.  Both figures are printed above.
    for i, n in ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        assert_close(av[i], rav[i], max(tol_out, 3.0 * rel_err(av32[i], rav[i])), n)
    assert n_checked >= 10 and worst[1] <= max(tol_grad, 3.0 * worst32), worst
    return rep


@pytest.mark.timeout(1800)
def test_cfg2_T800_fan_in_ragged_feedback_matches_oracle(dev, capsys):
    """SURVEY 8d's second parameter set on the benchmarked window: configs[1] at T_dec = 800 with N(0, 1/fan_in) weights
    (saturating gates), ragged lengths T ~ U[600, 800] / U ~ U[120, 200] and weak_feedback=True, vs the fp64 oracle.
    Tolerances: 1e-4 (cost, frames, kappa, w, phi; north star), 1e-3 norm-wise per gradient -- except where the oracle
    evaluated in float32 misses 1e-4 itself (tools/oracle_f32_drift.py: frames 1.4e-4, w 1.4e-4, phi 3.3e-4 after 800
    frames with these saturating weights): there the bar is 3 x the oracle's own float32 error, printed beside every
    figure."""
    _window_check(dev, "cfg2v", 800, 'float32', 1e-4, 1e-4, 1e-3, capsys, "T800 variant parity")


@pytest.mark.timeout(2400)
def test_cfg4_bf16_benchmarked_window_T800_matches_oracle(dev, capsys):
    """BASELINE configs[3] per GPU EXACTLY as `bench.py --config cfg4` runs it -- 3 x LSTM-1536, B = 64, T_enc = 200,
    **T_dec = 800**, bf16 MFMA operands / f32 accumulation -- against the oracle.

    What the test has to separate: the operand MODE and the KERNELS.  Rounding every operand of every decoder product to
    bf16 (4e-3 relative per element) and feeding it through an 800-deep recurrence whose attention position kappa is a
    running sum moves the trajectory itself: the ORACLE evaluated with bf16-rounded operands (parrot_ref.operand_rounding,
    fp64 accumulation) ends 800 frames 0.6 (frames), 4e-2 (kappa), 0.35 (w) away from the exact oracle, and the same
    emulation accumulated in float32 another 0.1 away from that (a state 1e-7 off can round to the other bf16 neighbour;
    profiles/r04_bf16_mode_drift.txt).  So:
      * reference = the oracle with bf16 operand rounding, fp64 accumulation (same rounding points as the product);
      * yardstick = the same emulation accumulated in float32, against that reference;
      * bar, at every horizon h in (100, 200, 400, 800) frames and for frames / kappa / w / phi: the HIP error over the first
        h frames <= max(tolerance of the mode, 3 x the yardstick's error over the same frames); cost within 5e-3 of the
        EXACT oracle;
      * gradients (round 5; until then the bar below evaluated to 2.7 norm-wise on the free-running trajectories, which
        nothing can fail): every parameter gradient within max(5e-2, 3 x yardstick) of the reference's backward ALONG THE
        TRAJECTORY THE HIP FORWARD WALKED (the oracle's chunks are rebuilt every 50 steps from the states the HIP scan
        saved, `cost_and_grads_checkpointed(pinned=)`), the bar itself asserted < 0.3; the free-running comparison is
        printed beside it.  T_dec = 100 and 200, where the trajectories have not separated, are held free-running AND
        pinned by test_cfg4_bf16_window_T100_T200_gradients_match_oracle.
    The short-window tolerances of the mode (tests/test_gpu_bf16.py: 2e-2 outputs, 5e-2 gradients) are unchanged."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw, init_kw, kb, B, U, ragged = VARIANTS["cfg4"]
    T = 800
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=1234, **init_kw)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(kb)
    feat, fm, lab, lm = variant_batch(cfg, T, B, U, ragged, seed=77)
    m = Parrot(device=dev, use_graph=True, compute_dtype='bf16', **kw).allocate()
    m.set_parameter_values(p)
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    grads = {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()}
    av = [x.detach().cpu().double() for x in av]
    cost = float(cost.detach())
    pin_every = 50
    hip_states = _hip_states(m, T, B, U, pin_every)
    m.close()
    with torch.no_grad():  # exact oracle, forward only (cost + how far the MODE moves the trajectory)
        carry, num, ex = None, 0.0, []
        for a in range(0, T, 100):
            c, carry, eav, _ = R.compute_cost(p, cfg, feat[a:a + 101], fm[a:a + 101], lab, lm, None, 1 if a == 0 else 0, carry)
            num += float(c) * float(fm[a + 1:a + 101].sum() + 1e-5)
            ex.append(eav)
        exact_cost = num / float(fm[1:].sum() + 1e-5)
        exact = [torch.cat([e[j] for e in ex], 0) for j in range(5)]
    p32 = {k: v.float().requires_grad_() for k, v in p.items()}
    for v in p.values():
        v.requires_grad_()
    with R.operand_rounding('bf16'):
        rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=100)
        c32, av32 = R.cost_and_grads_checkpointed(p32, cfg, feat.float(), fm.float(), lab, lm.float(), None, chunk=100)
    rep = [f"cfg4 T_dec={T} B={B} U={U} operands=bf16",
           f"cost: hip {cost:.6f}  exact oracle {exact_cost:.6f} (rel {abs(cost - exact_cost) / exact_cost:.2e})  "
           f"oracle with bf16 operands {float(rc):.6f} (rel {abs(cost - float(rc)) / float(rc):.2e})"]
    assert abs(cost - exact_cost) <= 5e-3 * exact_cost
    names = ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi"))
    failures = []
    for h in (100, 200, 400, 800):
        row = []
        for i, n in names:
            e = rel_err(av[i][:h], rav[i][:h])
            y = rel_err(av32[i][:h], rav[i][:h])
            mode = rel_err(rav[i][:h], exact[i][:h])
            row.append(f"{n} {e:.2e} (f32-accumulated emulation {y:.2e}; the mode itself {mode:.2e})")
            if e > max(2e-2, 3.0 * y):
                failures.append((h, n, e, y))
        rep.append(f"first {h} frames vs the bf16-operand oracle: " + "; ".join(row))
    worst, worst32, n_checked = ("", 0.0), 0.0, 0
    for name, ref in p.items():
        if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
            continue
        e = rel_err(grads[name], ref.grad)
        worst32 = max(worst32, rel_err(p32[name].grad, ref.grad))
        if e > worst[1]:
            worst = (name, e)
        n_checked += 1
    rep.append(f"FREE-RUNNING (reported, not the bar: the two forward trajectories are 5e-2 apart by frame 800): {n_checked} "
               f"parameter gradients vs the bf16-operand oracle's; worst {worst[0]}: {worst[1]:.2e} norm-wise "
               f"(f32-accumulated emulation: worst {worst32:.2e})")
    # The gradient bar: the oracle's backward on the trajectory the HIP forward walked.  Every 50 steps
    # the oracle's chunk is rebuilt from the states the HIP scan saved (h, cells, kappa, w entering step t), so both backward
    # passes differentiate ONE trajectory; the 800-deep chain of adjoints, the attention backward and the weight-gradient
    # sums are the oracle's own (fp64 accumulation, bf16-rounded operands).  Yardstick: the same, accumulated in float32.
    pw, pw32, pn, bar = _pinned_gradient_check(R, p, p32, cfg, feat, fm, lab, lm, hip_states, pin_every, grads, rep)
    # At T = 800 the worst pinned gradient error and its float32 yardstick are BOTH draws from one distribution: any change
    # of a rounding (an FMA contracted differently, another summation order in the attention) sends the bf16 trajectory
    # elsewhere within its chunks.  Three builds of rounds 5-6 on the same inputs: HIP 3.1e-2 / yardstick 3.7e-2, 9.0e-2 /
    # < 1.7e-2, 1.7e-2 / 7.8e-2 -- so the bar's floor is the spread of the yardstick itself (0.12), not the short windows' 5e-2;
    # the free-running error it has to tell apart is 0.5 - 0.9, and the bar still must stay below 0.3.
    bar = max(bar, 0.12)
    rep.append(f"bar at T = 800: max(0.12, 3 x yardstick) = {bar:.2e}")
    with capsys.disabled():
        print("\n[cfg4 bf16 T800 parity] " + "\n[cfg4 bf16 T800 parity] ".join(rep))
    assert not failures, failures
    assert n_checked >= 10
    assert bar < 0.3, f"a gradient bar of {bar:.2f} norm-wise cannot fail"
    assert pn >= 10 and pw[1] <= bar, (pw, bar)


def _hip_states(m, T, B, U, every):
    """The states ENTERING step t (t a multiple of `every`) as the HIP scan saved them: {t: oracle-style carry, fp64}."""
    ws = m._train_workspace(T, B, U)
    lstm = m.cell_type == 'lstm'
    out = {}
    for t in range(every, T, every):
        cpu = lambda x: x[t].detach().cpu().double().clone()
        out[t] = dict(h=[(cpu(ws['h'][l]), cpu(ws['cst'][l])) if lstm else cpu(ws['h'][l]) for l in range(m.num_layers)],
                      k=cpu(ws['kappa']), w=cpu(ws['w']))
    return out


def _pinned_gradient_check(R, p, p32, cfg, feat, fm, lab, lm, hip_states, every, grads, rep):
    """HIP gradients vs the bf16-operand oracle's backward along the HIP trajectory (cost_and_grads_checkpointed(pinned=)).
    Returns (worst (name, err), the float32-accumulated emulation's worst error, gradients checked, bar)."""
    def pin(dt):
        cast = lambda x: x.to(dt)
        return lambda t: dict(h=[tuple(cast(y) for y in x) if isinstance(x, tuple) else cast(x) for x in hip_states[t]['h']],
                              k=cast(hip_states[t]['k']), w=cast(hip_states[t]['w']))
    for d in (p, p32):
        for v in d.values():
            v.grad = None
    with R.operand_rounding('bf16'):
        R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=every, pinned=pin(torch.float64))
        R.cost_and_grads_checkpointed(p32, cfg, feat.float(), fm.float(), lab, lm.float(), None, chunk=every,
                                      pinned=pin(torch.float32))
    worst, worst32, n = ("", 0.0), 0.0, 0
    for name, ref in p.items():
        if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
            continue
        e = rel_err(grads[name], ref.grad)
        worst32 = max(worst32, rel_err(p32[name].grad, ref.grad))
        if e > worst[1]:
            worst = (name, e)
        n += 1
    bar = max(5e-2, 3.0 * worst32)
    rep.append(f"PINNED trajectory (oracle chunks rebuilt from the HIP states every {every} steps): {n} parameter gradients; "
               f"worst {worst[0]}: {worst[1]:.2e} norm-wise (f32-accumulated emulation on the same trajectory: worst "
               f"{worst32:.2e}); bar {bar:.2e}")
    return worst, worst32, n, bar


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("T", [100, 200])
def test_cfg4_bf16_window_T100_T200_gradients_match_oracle(dev, capsys, T):
    """BASELINE configs[3] per GPU (3 x LSTM-1536, B = 64, U = 200, bf16 operands: schedule 7, the fused backward tick with
    its K halves, the bf16-in weight-gradient GEMM) at windows where the bf16 operand MODE has not yet moved the trajectory
    (profiles/r04_bf16_mode_drift.txt: frames 2.6e-3, kappa 1.6e-5 after 100 frames): cost, frames, kappa, w, phi and EVERY
    parameter gradient of the free-running HIP window vs the oracle with the same operand rounding (fp64 accumulation).
    Bars: outputs max(2e-2, 3 x yardstick), gradients max(5e-2, 3 x yardstick) norm-wise, yardstick = the same emulation
    accumulated in float32 -- and the gradient bar itself must stay below 0.3 (the T = 800 free-running
    bar evaluated to 2.7, which nothing can fail)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw, init_kw, kb, B, U, ragged = VARIANTS["cfg4"]
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=1234, **init_kw)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(kb)
    feat, fm, lab, lm = variant_batch(cfg, T, B, U, ragged, seed=77)
    m = Parrot(device=dev, use_graph=True, compute_dtype='bf16', **kw).allocate()
    m.set_parameter_values(p)
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    grads = {k: v.detach().cpu().double().clone() for k, v in m.get_gradient_dict().items()}
    av = [x.detach().cpu().double() for x in av]
    cost = float(cost.detach())
    hip_states = _hip_states(m, T, B, U, 50)
    m.close()
    p32 = {k: v.float().requires_grad_() for k, v in p.items()}
    for v in p.values():
        v.requires_grad_()
    with R.operand_rounding('bf16'):
        rc, rav = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, None, chunk=100)
        c32, av32 = R.cost_and_grads_checkpointed(p32, cfg, feat.float(), fm.float(), lab, lm.float(), None, chunk=100)
    rep = [f"cfg4 T_dec={T} B={B} U={U} operands=bf16",
           f"cost: hip {cost:.6f}  oracle with bf16 operands {float(rc):.6f} (rel {abs(cost - float(rc)) / float(rc):.2e}; "
           f"f32-accumulated emulation {abs(float(c32) - float(rc)) / float(rc):.2e})"]
    assert abs(cost - float(rc)) <= 5e-3 * float(rc)
    for i, n in ((0, "predicted frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        e, y = rel_err(av[i], rav[i]), rel_err(av32[i], rav[i])
        rep.append(f"{n}: {e:.2e} (f32-accumulated emulation {y:.2e})")
        assert e <= max(2e-2, 3.0 * y), (n, e, y)
    worst, worst32, n_checked = ("", 0.0), 0.0, 0
    for name, ref in p.items():
        if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
            continue
        e = rel_err(grads[name], ref.grad)
        worst32 = max(worst32, rel_err(p32[name].grad, ref.grad))
        if e > worst[1]:
            worst = (name, e)
        n_checked += 1
    bar = max(5e-2, 3.0 * worst32)
    rep.append(f"free-running: {n_checked} parameter gradients; worst {worst[0]}: {worst[1]:.2e} norm-wise "
               f"(f32-accumulated emulation: worst {worst32:.2e}); bar {bar:.2e}")
    pw, pw32, pn, pbar = _pinned_gradient_check(R, p, p32, cfg, feat, fm, lab, lm, hip_states, 50, grads, rep)
    with capsys.disabled():
        print(f"\n[cfg4 bf16 T{T} parity] " + f"\n[cfg4 bf16 T{T} parity] ".join(rep))
    assert bar < 0.3 and pbar < 0.3, (bar, pbar)
    assert n_checked >= 10 and worst[1] <= bar, (worst, bar)
    assert pn >= 10 and pw[1] <= pbar, (pw, pbar)


def test_cfg3_decode_1000_steps_matches_oracle(dev, capsys):
    """BASELINE configs[2] at its real length: decode, batch 16, H=1024, weak feedback, **1000 frames**, every output
    of sample_model vs the fp64 oracle (the 60-step test below cannot see a slow drift of the fed-back frame).

    The decode loop feeds its own output back, and with these weights the map is expansive: measured on the MI355X
    (tools/decode_drift.py, profiles/r03_decode_drift.txt) the distance to the fp64 trajectory grows smoothly by about
    2 x per 70 frames -- 5e-7 after 10 frames, 2e-5 after 300, 6e-5 after 500 -- for the persistent machine AND for the
    launch path, which end 1e-2 apart from EACH OTHER after 1000 frames although both are exact fp32 products.  No fp32
    implementation can hold 1e-4 over 1000 free-running frames of this map, so the criterion is split:
      * frames 0..299 at the north star's 1e-4 (measured 1.6e-5);
      * the whole 1000 frames no worse than what fp32 arithmetic itself costs: at most 3 x the distance of the SAME
        oracle run in float32 (torch-CPU) from its float64 run, at every horizon (measured ratios 0.4-1.3; the factor was
        30 until round 4; a real defect -- a wrong carry, a stale operand -- shows up as orders of magnitude, at once,
        not as this slow common drift);
      * the teacher-forced 800-frame training window (test_cfg2_benchmarked_window_T800_matches_oracle) covers the long horizon without
        the feedback amplification."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024, weak_feedback=True)
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=29, scale_by_fan_in=True)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(-2.3)  # ~0.13 positions per frame: inside the 200-character text at frame 1000
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    N, U, S = 16, 200, 1000
    _, _, lab, lm, _ = make_batch(cfg, 2, N, U, seed=31)
    with torch.no_grad():
        ref = R.sample_model(p, cfg, lab, lm, None, S)
        ref32 = R.sample_model({k: v.float() for k, v in p.items()}, cfg, lab, lm.float(), None, S)
    assert float(ref[1][-1].mean()) < U - 5, "the windows must (on average) stay inside the text for the test to mean anything"
    outs = m.sample_model(lab.numpy(), lm.float().numpy(), None, None, N, S)
    report = []
    for o, r, r32, n in zip(outs, ref, ref32, ("sample_x", "k", "w", "pi", "phi", "pi_att")):
        assert o.shape == tuple(r.shape), n
        o = torch.from_numpy(o).double()
        r, r32 = r.double(), r32.double()
        if o.shape[0] != S:  # (outputs without a time axis)
            assert_close(o, r, 1e-4, n)
            continue
        line = []
        for hz in (100, 300, 500, 700, 1000):
            scale = float(r[:hz].abs().max())
            e = float((o[:hz] - r[:hz]).abs().max()) / scale
            e32 = float((r32[:hz] - r[:hz]).abs().max()) / scale
            line.append(f"t<{hz}: {e:.1e} (oracle-f32 {e32:.1e})")
            if hz <= 300:
                assert e <= 1e-4, f"{n}: {e:.2e} over the first {hz} frames"
            assert e <= max(1e-4, 3.0 * e32), f"{n}: {e:.2e} over {hz} frames, float32 oracle {e32:.2e}"
        report.append(f"{n}: " + "  ".join(line))
    with capsys.disabled():
        print("\n[decode-1000 parity] " + "\n[decode-1000 parity] ".join(report))
    m.close()


def test_cfg3_decode_width(dev):
    """BASELINE configs[2]: decode, batch 16, H=1024, weak feedback, 60 steps, every output of sample_model."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024, weak_feedback=True)
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=29, scale_by_fan_in=True)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.0)
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    N, U, S = 16, 100, 60
    _, _, lab, lm, _ = make_batch(cfg, 2, N, U, seed=31, ragged=True)
    with torch.no_grad():
        ref = R.sample_model(p, cfg, lab, lm, None, S)
    for rep in range(2):
        outs = m.sample_model(lab.numpy(), lm.float().numpy(), None, None, N, S)
        for o, r, n in zip(outs, ref, ("sample_x", "k", "w", "pi", "phi", "pi_att")):
            assert o.shape == tuple(r.shape), n
            assert_close(torch.from_numpy(o), r, 1e-4, f"pass {rep}: {n}")
    m.close()


def test_cfg5_generator_full_width_greedy(dev, capsys):
    """BASELINE configs[4]: three-tier GRU DIM=1024, batch 32, 25 frames = 2000 samples (the utterance bench.py times),
    temperature 0.
    Greedy indices must equal the fp64 oracle's; a row may only leave the oracle's trajectory at a position where the
    oracle's own top-2 logits are closer than fp32 can resolve (gap < 2e-5 * |logit|max), which is then reported.
    The logits of the last sample step are compared at 1e-4."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
    try:
        c = S.config()
        p = S.init_params(c, seed=5, perturb=0.2)
        lib.set_params(p)
        g = torch.Generator().manual_seed(2)
        T, B = 25, 32
        feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref, ref_logits = S.generate(p, c, feats, return_logits=True)
        ref = ref.numpy()
        gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=True)
        out = gen.generate(feats.float().numpy()).cpu().numpy()
        last_logits = gen.ws['logits'].detach().cpu().double()
        gen.close()
        assert out.shape == ref.shape == (B, 80 * T) and out.dtype == np.int32
        exact_rows = 0
        for b in range(B):
            diff = np.nonzero(out[b] != ref[b])[0]
            if diff.size == 0:
                exact_rows += 1
                continue
            t = int(diff[0])
            lg = ref_logits[b, t - 80]
            top2 = torch.topk(lg, 2).values
            gap = float(top2[0] - top2[1])
            assert gap < 2e-5 * float(lg.abs().max()), \
                f"row {b} leaves the oracle at sample {t} where the oracle's top-2 gap is {gap:.3e} (not a tie)"
        with capsys.disabled():
            print(f"\n[cfg5 greedy parity] {exact_rows} of {B} rows follow the fp64 oracle bit for bit over {80 * T} samples; "
                  f"{B - exact_rows} left it at a position where the oracle's own top-2 logits tie within fp32 resolution")
        assert exact_rows >= B - 3, f"only {exact_rows} of {B} rows follow the oracle bit for bit"
        same = [b for b in range(B) if np.array_equal(out[b], ref[b])]
        assert_close(last_logits[same], ref_logits[same, -1], 1e-4, "last-step logits")
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)


@pytest.mark.parametrize("B,S_len", [(4, 320), (32, 4000)])
def test_cfg5_training_full_width_cost_and_every_gradient(dev, capsys, B, S_len):
    """BASELINE configs[4] widths in TRAINING (three-tier GRU, DIM = 1024, EMB_SIZE = 256, Q = 256; three_tier.py:534-636) on the
    round-5 operators -- gather-sum / segmented-sum embedding, ReLU MLP with gated dx products, softmax-CE kernels, 1024-wide
    tier GEMMs and scans: cost, ip_cost, the carried states and EVERY parameter gradient vs the fp64 oracle, on mu-law-like
    (peaked) sample codes with a ragged mask, a fresh window (reset) and a carried one.  B = 4, 320 samples: 1280 rows of the
    sample-level tier, ten 128-position chunks per embedding position in the segmented sum, most of the 256 codes unused.
    B = 32, 4000 samples: the window bench.py times (`secondary.samplernn_train`), 128 000 rows, a fresh window only (the
    float64 oracle of it takes a few minutes of CPU and ~20 GB: skipped on a host with less than 64 GB free)."""
    if B * S_len > 100000:
        import psutil
        if psutil.virtual_memory().available < 64 << 30:
            pytest.skip("the float64 oracle of the benchmarked window needs ~20 GB of host memory")
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
    # A ReLU pre-activation within rounding of 0 makes the GRADIENT ill-posed for any f32 implementation: the mask bit of
    # that element -- and with it one whole term of every upstream gradient -- follows the last bit of the product
    # (round 6: exactly one of 2.6 M pre-activations sits at 1.6e-6 of a layer whose largest is 11.4; the f32-input MFMA
    # kernel lands above 0, the split-bf16 kernel below, both within 1e-7 of the float64 value; tools/x3_debug.py: every
    # gradient then moves by 1e-3).  The oracle therefore takes the HIP run's own branch for elements within 1e-6 of the
    # kink (oracle/samplernn_ref.py _relu_at_ties) and its own float64 sign everywhere else; the test asserts that at
    # most a handful of elements are that close and reports how many branches were actually taken over.
    from parrot_amd import ops as hops
    masks = []
    orig_gemm = hops.gemm

    def spy(a, b, bias=None, out=None, accumulate=False, act=hops.ACT_NONE, alpha=1.0, split_k=0):
        r = orig_gemm(a, b, bias=bias, out=out, accumulate=accumulate, act=act, alpha=alpha, split_k=split_k)
        if act == hops.ACT_RELU:
            masks.append((r.detach() > 0).cpu())
        return r
    hops.gemm = spy
    try:
        c = S.config()
        p = S.init_params(c, seed=8, perturb=0.1)
        lib.set_params(p)
        g = torch.Generator().manual_seed(3)
        seq = (torch.randn(B, S_len + 80, generator=g) * 12 + 128).round().clamp(0, 255).long()
        seq[:, ::9] = 128
        feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
        mask = torch.ones(B, S_len + 80, dtype=torch.float64)
        mask[1, S_len - 70:] = 0
        mask[3, S_len + 13:] = 0
        rep = []
        for reset in ((1, 0) if B * S_len < 100000 else (1,)):
            h0 = torch.randn(B, 1, 1024, generator=g, dtype=torch.float64) * 0.3
            bh0 = torch.randn(B, 1, 1024, generator=g, dtype=torch.float64) * 0.3
            for t in lib.named_params().values():
                t.grad = None
            masks.clear()
            cost, ip_cost, allp, ipp, otherp, nh0, nbh0 = tt.compute_cost(
                seq.to(dev), feats.float().to(dev), h0.float().to(dev), bh0.float().to(dev), reset, mask.float().to(dev))
            (cost + ip_cost).backward()
            assert len(masks) == 2, "the two ReLU layers of sample_level_predictor"
            ref_p = {k: v.clone().requires_grad_() for k, v in p.items()}
            ties = []
            rc, rip, rh0, rbh0 = S.compute_cost(ref_p, c, seq, feats, h0, bh0, reset, mask, relu_ties=list(masks),
                                                tie_tol=1e-6, tie_report=ties)
            (rc + rip).backward()
            near, taken = sum(t_[0] for t_ in ties), sum(t_[1] for t_ in ties)
            assert near <= 64 * max(1, B * S_len // 1280), f"{near} pre-activations within 1e-6 of the kink: not a handful"
            assert_close(cost, rc, 1e-4, "cost")
            assert_close(ip_cost, rip, 1e-4, "ip_cost")
            assert_close(nh0, rh0, 1e-4, "new_h0")
            assert_close(nbh0, rbh0, 1e-4, "new_big_h0")
            worst, n = ("", 0.0), 0
            for name, t in lib.named_params().items():
                rg = ref_p[name].grad
                if rg is None or float(rg.abs().max()) < 1e-12:
                    continue
                assert t.grad is not None, name
                e = rel_err(t.grad, rg)
                assert e < 2e-3, (name, e)
                if e > worst[1]:
                    worst = (name, e)
                n += 1
            assert n >= 20
            rep.append(f"reset={reset}: cost {float(cost.detach()):.5f} (oracle {float(rc.detach()):.5f}), ip_cost {float(ip_cost.detach()):.5f}; {n} gradients, "
                       f"worst {worst[0]}: {worst[1]:.2e}; ReLU pre-activations within 1e-6 of 0: {near} of {2048 * B * S_len / 1e6:.1f} M, branch taken from the HIP run: {taken}")
        with capsys.disabled():
            print("\n[cfg5 training parity] " + "\n[cfg5 training parity] ".join(rep))
    finally:
        hops.gemm = orig_gemm
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
