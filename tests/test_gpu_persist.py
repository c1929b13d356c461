"""The persistent phase machine (PARROT_SCHEDULE=4, parrot_amd/csrc/persist.hip): the forward scan of
Parrot.compute_cost as ONE resident kernel with LDS-stationary weights.  Parity against the oracle -- cost, frames,
window state and every gradient (the backward runs on the launch schedule from the buffers the machine saved) -- plus
run-to-run bitwise reproducibility, hipGraph replay, TBPTT carry and the abort word."""
import pytest
import torch

from tests.util import assert_close, make_batch, rel_err

pytestmark = pytest.mark.gpu

SMALL = dict(rnn_h_dim=64, readouts_dim=48, encoder_dim=16, input_dim=24, speaker_dim=8, num_speakers=5,
             encoder_type='bidirectional')


def _is_persistent(m, T, B, U):
    from parrot_amd import _lib
    ws = m._train_ws.get(('dec', T, B, U))
    assert ws is not None
    if 'persist_ws' in ws:
        assert int(ws['persist_ws'][832:833].view(torch.int32).item()) == 0, "a spin timed out inside the machine"
    return bool(_lib.load().parrot_decoder_is_persistent(ws['plan']))


def _check(dev, monkeypatch, T, B, U, kw, use_graph=True, expect_persistent=True, tol_grad=1e-3, seed=3):
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    monkeypatch.setenv("PARROT_SCHEDULE", "4")
    full = dict(SMALL, **kw)
    cfg = R.default_config(**full)
    p = R.init_params(cfg, seed=7, scale_by_fan_in=True)
    m = Parrot(device=dev, use_graph=use_graph, **full).allocate()
    m.set_parameter_values(p)
    feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=seed, ragged=True, speaker=cfg['use_speaker'])
    for v in p.values():
        v.requires_grad_()
    rc, _, rav, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, spk, 1)
    rc.backward()
    outs = []
    for rep in range(2):
        m.zero_grad()
        cost, upd, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev),
                                          None if spk is None else spk.to(dev), 1, B)
        cost.backward()
        assert _is_persistent(m, T, B, U) == expect_persistent
        assert_close(cost, rc, 1e-4, "cost")
        assert_close(av[0], rav[0], 1e-4, "frames")
        assert_close(av[1], rav[1], 1e-4, "kappa")
        assert_close(av[2], rav[2], 1e-4, "w")
        assert_close(av[4], rav[4], 1e-4, "phi")
        assert_close(av[5], rav[5], 1e-4, "pi_att")
        grads = m.get_gradient_dict()
        for name, ref in p.items():
            if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
                continue
            e = rel_err(grads[name], ref.grad)
            assert e <= tol_grad, f"grad {name}: rel err {e:.3e}"
        outs.append((cost.detach().clone(), av[0].clone(), av[1].clone(), m.flat_gradients.clone()))
    # no float atomics, fixed reduction orders: two passes (the second replays the captured graph) agree bit for bit
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    m.close()


@pytest.mark.parametrize("L,B", [(1, 5), (2, 20), (2, 64), (3, 37), (3, 64)])
def test_persistent_forward_layers_and_row_blocks(dev, monkeypatch, L, B):
    """1, 2 and 4 row blocks (B <= 16 / 32 / 64), 1-3 layers; T = 9 covers the lagging layers' fill and drain."""
    _check(dev, monkeypatch, T=9, B=B, U=11, kw=dict(num_layers=L))


def test_persistent_forward_feedback_speaker_softmax(dev, monkeypatch):
    _check(dev, monkeypatch, T=8, B=6, U=9, kw=dict(num_layers=3, full_feedback=True, use_speaker=True))
    _check(dev, monkeypatch, T=7, B=5, U=8, kw=dict(num_layers=2, weak_feedback=True, attention_type='softmax'),
           use_graph=False)


def test_non_qualifying_configurations_fall_back(dev, monkeypatch):
    """LSTM layers / layer_norm / B > 64 are not covered by the machine: the launch schedules run instead."""
    _check(dev, monkeypatch, T=6, B=4, U=7, kw=dict(num_layers=2, cell_type='lstm'), expect_persistent=False)
    _check(dev, monkeypatch, T=6, B=4, U=7, kw=dict(num_layers=2, layer_norm=True, weak_feedback=True),
           expect_persistent=False, tol_grad=2e-3)
    _check(dev, monkeypatch, T=4, B=70, U=7, kw=dict(num_layers=2), expect_persistent=False)


def test_persistent_forward_tbptt_carry_and_launch_path_agreement(dev, monkeypatch):
    """One window == two windows with the carried state; and the machine's frames agree with the launch schedule's to
    rounding (different summation order of the same products)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(SMALL, num_layers=2, weak_feedback=True)
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=7, scale_by_fan_in=True)
    T, B, U = 12, 9, 7
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=5)
    args = [t.to(dev) for t in (feat.float(), fm.float(), lab, lm.float())]
    res = {}
    for sched in ("0", "4"):
        monkeypatch.setenv("PARROT_SCHEDULE", sched)
        m = Parrot(device=dev, use_graph=True, **kw).allocate()
        m.set_parameter_values(p)
        c, upd, av, _ = m.compute_cost(args[0], args[1], args[2], args[3], None, 1, B)
        full = av[0].clone()
        c1, u1, av1, _ = m.compute_cost(args[0][:7], args[1][:7], args[2], args[3], None, 1, B)
        m.apply_updates(u1)
        c2, u2, av2, _ = m.compute_cost(args[0][6:], args[1][6:], args[2], args[3], None, 0, B)
        assert_close(torch.cat([av1[0], av2[0]], 0), full, 1e-5, f"schedule {sched}: TBPTT carry")
        res[sched] = full
        m.close()
    assert_close(res["4"], res["0"], 1e-5, "machine vs launches")


def test_persistent_forward_cfg2_width(dev, monkeypatch):
    """BASELINE configs[1] widths (H = R = 1024, B = 64, U = 200): LDS-resident and streamed units, 4 row blocks."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    monkeypatch.setenv("PARROT_SCHEDULE", "4")
    kw = dict(num_layers=2, encoder_type='bidirectional', rnn_h_dim=1024, readouts_dim=1024)
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, seed=11, scale_by_fan_in=True)
    p['/parrot/h1_to_att/fork_kappa.b'].fill_(-1.5)
    m = Parrot(device=dev, use_graph=True, **kw).allocate()
    m.set_parameter_values(p)
    T, B, U = 6, 64, 200
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=12, ragged=True)
    for v in p.values():
        v.requires_grad_()
    rc, _, rav, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, None, 1)
    rc.backward()
    m.zero_grad()
    cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
    cost.backward()
    assert _is_persistent(m, T, B, U)
    assert_close(cost, rc, 1e-4, "cost")
    for i, n in ((0, "frames"), (1, "kappa"), (2, "w"), (4, "phi")):
        assert_close(av[i], rav[i], 1e-4, n)
    grads = m.get_gradient_dict()
    for name, ref in p.items():
        if ref.grad is None or float(ref.grad.abs().max()) < 1e-12:
            continue
        assert rel_err(grads[name], ref.grad) <= 1e-3, name
    m.close()


def test_workspaces_and_plans_are_reused_between_calls(dev):
    """A second compute_cost / sample_model call with the same shapes must reuse the cached workspace and replay its
    instantiated hipGraph (round 1 stored the cache entry under a shadowed key and rebuilt both on every call)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    kw = dict(SMALL, num_layers=2, weak_feedback=True)
    cfg = R.default_config(**kw)
    m = Parrot(device=dev, use_graph=True, **kw).initialize()
    feat, fm, lab, lm, _ = make_batch(cfg, 6, 4, 7, seed=5)
    args = (feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev))
    plans = []
    for _ in range(3):
        c, upd, av, _ = m.compute_cost(*args, None, 1, 4)
        c.backward()
        ws = m._train_ws.get(('dec', 6, 4, 7))
        assert ws is not None
        plans.append(ws['plan'].value)
    assert len(set(plans)) == 1 and len(m._train_ws) == 2  # encoder runner + decoder workspace
    for _ in range(2):
        m.sample_model_device(lab, lm.float(), None, 4, 5)
    assert len(m._sample_ws) == 1
    m.close()


@pytest.mark.parametrize("kw", [dict(num_layers=2, weak_feedback=True),
                                dict(num_layers=3, full_feedback=True, use_speaker=True),
                                dict(num_layers=1),
                                dict(num_layers=2, weak_feedback=True, sharpening_coeff=1.2, timing_coeff=0.9,
                                     attention_type='softmax')])
def test_decode_on_the_persistent_machine(dev, monkeypatch, kw):
    """sample_model as one resident kernel -- 2L + 2 phases per step with every product cut along K by the age of its
    operands, 2L + 1 with the fed-back frame out of the chain where that applies (default; PARROT_PM_FBC=0 for the former),
    or the 2L + 3 whole-K phases (PARROT_PM_PIECES=0): every output vs the oracle, the plan that was
    asked for really ran, a second call replays it, and the per-step launch path (PARROT_SAMPLE_PERSIST=0) agrees to
    rounding with both."""
    from oracle import parrot_ref as R
    from parrot_amd import _lib
    from parrot_amd.model import Parrot
    full = dict(SMALL, **kw)
    cfg = R.default_config(**full)
    p = R.init_params(cfg, seed=7, scale_by_fan_in=True)
    N, U, S = 5, 9, 14
    _, _, lab, lm, spk = make_batch(cfg, 2, N, U, seed=9, speaker=cfg['use_speaker'])
    with torch.no_grad():
        ref = R.sample_model(p, cfg, lab, lm, spk, S)
    res = {}
    fbc_applies = full['num_layers'] >= 2 and full.get('weak_feedback') and not full.get('full_feedback')
    for mode, pieces, fbc in (("1", "1", "1"), ("1", "1", "0"), ("1", "0", "1"), ("0", "1", "1")):
        if fbc == "0" and not fbc_applies:
            continue
        monkeypatch.setenv("PARROT_SAMPLE_PERSIST", mode)
        monkeypatch.setenv("PARROT_PM_PIECES", pieces)
        monkeypatch.setenv("PARROT_PM_FBC", fbc)
        m = Parrot(device=dev, use_graph=True, **full).allocate()
        m.set_parameter_values(p)
        for rep in range(2):
            outs = m.sample_model_device(lab, lm.float(), spk, N, S)
            for o, r, n in zip(outs, ref, ("sample_x", "k", "w", "pi", "phi", "pi_att")):
                assert_close(o, r, 1e-4, f"persist={mode} pass {rep}: {n}")
        ws = m._sample_ws.get((S, N, U))
        # 3: the step cut along K with the fed-back frame out of the chain (weak feedback, L >= 2; round 5), 2: cut along K
        want = 0 if mode == "0" else ((3 if (fbc == "1" and fbc_applies) else 2) if pieces == "1" else 1)
        assert _lib.load().parrot_sample_is_persistent(ws['plan']) == want
        if mode == "1":
            assert int(ws['pm']['ws'][832:833].view(torch.int32).item()) == 0, "a spin timed out inside the machine"
        res[mode + pieces + fbc] = [o.clone() for o in outs]
        m.close()
    for key in ("111", "110", "101"):
        if key not in res:
            continue
        for a, b, n in zip(res[key], res["011"], ("sample_x", "k", "w", "pi", "phi", "pi_att")):
            assert_close(a, b, 2e-5, f"machine ({key}) vs launches: {n}")


def test_dataflow_mode_matches_the_barrier_mode_bit_for_bit(dev, monkeypatch):
    """PARROT_PM_DATAFLOW=1: no grid barriers, consumers re-read 16-byte slots until they stop being EMPTY (persist.h).
    Same units, same arithmetic: the training forward (cost, frames) and the decode outputs equal the barrier mode's
    bit for bit, with padding rows (B = 5, 37), three layers, feedback, and a second call on the same workspace."""
    from oracle import parrot_ref as R
    from parrot_amd import _lib
    from parrot_amd.model import Parrot
    monkeypatch.setenv("PARROT_SCHEDULE", "4")
    for kw, T, B, U in ((dict(num_layers=3, weak_feedback=True), 9, 37, 11), (dict(num_layers=2), 8, 5, 9)):
        full = dict(SMALL, **kw)
        cfg = R.default_config(**full)
        p = R.init_params(cfg, seed=4, scale_by_fan_in=True)
        feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=5, ragged=True)
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("PARROT_PM_DATAFLOW", mode)
            m = Parrot(device=dev, **full).allocate()
            m.set_parameter_values(p)
            for rep in range(2):
                m.zero_grad()
                cost, _, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev),
                                                None, 1, B)
                cost.backward()
            assert _is_persistent(m, T, B, U)
            outs = m.sample_model_device(lab, lm.float(), None, B, 6)
            ws = m._train_workspace(T, B, U)
            _lib.call('parrot_decoder_status', ws['plan'])  # raises if a launch gave up
            got[mode] = [cost.detach().clone(), av[0].clone(), av[2].clone()] + [o.clone() for o in outs[:3]]
            m.close()
        for a, b in zip(got["0"], got["1"]):
            assert torch.equal(a, b)
