"""CPU tests: the oracle against the reference's golden vectors / known answers, and basic
self-consistency of the restatements (no GPU, no HIP compute)."""
import os

import numpy as np
import torch

from oracle import parrot_ref as R
from oracle import quantize_ref as Q

GOLD = os.path.join(os.path.dirname(__file__), "golden", "quantize_golden.npz")


def test_quantize_oracle_matches_reference_golden():
    g = np.load(GOLD)
    mu = Q.batch_quantize(g["x"], 256, "mu-law")
    lin = Q.batch_quantize(g["x"], 256, "linear")
    assert mu.dtype == np.int16 and np.array_equal(mu, g["mu"])
    assert lin.dtype == np.int32 and np.array_equal(lin, g["lin"])
    dec = Q.mu2linear(g["mu"])
    assert dec.dtype == np.float32 and np.array_equal(dec, g["dec"])


def test_quantize_docstring_known_answers():
    """quantize.py:55-63: encode range 0..255 int16, decode range -1..0.9574371 float32."""
    out = Q.linear2mu(np.array([[-1.0, 0.0, 1.0]]))
    assert out.dtype == np.int16 and out.tolist() == [[0, 127, 255]]
    dec = Q.mu2linear(np.array([0, 255], dtype=np.int16))
    assert dec.dtype == np.float32
    assert dec[0] == -1.0 and abs(float(dec[1]) - 0.9574371) < 1e-7
    g = np.load(GOLD)
    assert np.array_equal(g["kat_mu"], out) and np.array_equal(g["kat_dec"], dec)


def test_quantize_rows_are_independent_and_cover_range():
    rng = np.random.RandomState(0)
    x = rng.randn(5, 999).astype(np.float32)
    q = Q.batch_quantize(x, 256, "mu-law")
    assert q.min(axis=1).tolist() == [0] * 5 and q.max(axis=1).tolist() == [255] * 5
    q1 = Q.batch_quantize(x[2:3], 256, "mu-law")
    assert np.array_equal(q1[0], q[2])


def _tiny(**kw):
    cfg = R.default_config(rnn_h_dim=16, readouts_dim=12, encoder_type='bidirectional', encoder_dim=4,
                           input_dim=6, **kw)
    return cfg, R.init_params(cfg, seed=3, scale_by_fan_in=True)


def test_tbptt_carry_equivalence():
    """One long window == two short windows with carried state (start_flag=0), SURVEY section 4."""
    cfg, p = _tiny(num_layers=3, weak_feedback=True)
    from tests.util import make_batch
    T, B, U = 8, 3, 5
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=1)
    _, _, av, ex = R.compute_cost(p, cfg, feat, fm, lab, lm, None, 1)
    c1, carry, av1, _ = R.compute_cost(p, cfg, feat[:5], fm[:5], lab, lm, None, 1)
    c2, _, av2, _ = R.compute_cost(p, cfg, feat[4:], fm[4:], lab, lm, None, 0, carry=carry)
    both = torch.cat([av1[0], av2[0]], 0)
    assert torch.allclose(both, av[0], atol=1e-12)


def test_checkpointed_bptt_equals_one_piece_backward():
    """R.cost_and_grads_checkpointed (what the T_dec = 800 GPU parity test uses to fit the host memory) gives the cost,
    the outputs and EVERY parameter gradient of the one-piece compute_cost(...).backward(): ragged masks, feedback,
    speaker, 3 layers, LSTM layers (the cfg4 T_dec = 800 test), a ragged last chunk."""
    from tests.util import make_batch
    for kw in (dict(num_layers=3, weak_feedback=True, use_speaker=True, num_speakers=4, speaker_dim=5),
               dict(num_layers=2), dict(num_layers=3, cell_type='lstm')):  # LSTM layers carry (state, cells) pairs
        cfg, p = _tiny(**kw)
        T, B, U = 11, 3, 6
        feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=5, ragged=True, speaker=cfg['use_speaker'])
        for v in p.values():
            v.requires_grad_()
        c, _, av, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, spk, 1)
        c.backward()
        ref = {k: v.grad.clone() for k, v in p.items() if v.grad is not None}
        for v in p.values():
            v.grad = None
        c2, av2 = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, spk, chunk=4)
        assert abs(float(c2) - float(c)) < 1e-12 * abs(float(c))
        for x, y in zip(av, av2):
            assert torch.allclose(x.detach(), y, atol=1e-12)
        assert set(ref) == {k for k, v in p.items() if v.grad is not None}
        for k, g in ref.items():
            assert torch.allclose(p[k].grad, g, rtol=1e-10, atol=1e-13), k


def test_checkpointed_bptt_pinned_to_its_own_trajectory_changes_nothing():
    """cost_and_grads_checkpointed(pinned=...) rebuilds every chunk from a state handed in from outside (the GPU tests
    hand in the HIP forward's saved states so that a long bf16 window's backward is judged on ONE trajectory).  Pinned to
    the oracle's own trajectory it must reproduce the unpinned gradients; pinned to a perturbed one it must not."""
    from tests.util import make_batch
    for kw in (dict(num_layers=2), dict(num_layers=3, cell_type='lstm')):
        cfg, p = _tiny(**kw)
        T, B, U, chunk = 11, 3, 6, 4
        feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=5, ragged=True)
        for v in p.values():
            v.requires_grad_()
        c, av = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, spk, chunk=chunk)
        ref = {k: v.grad.clone() for k, v in p.items() if v.grad is not None}
        with torch.no_grad():  # the states entering every step, from a one-piece forward
            carry, states = None, {}
            for t in range(T):
                _, carry, _, _ = R.compute_cost(p, cfg, feat[t:t + 2], fm[t:t + 2], lab, lm, spk, 1 if t == 0 else 0, carry)
                states[t + 1] = carry
        for scale, same in ((0.0, True), (1e-3, False)):
            for v in p.values():
                v.grad = None

            def pinned(t):
                bump = lambda x: x * (1.0 + scale)
                s = states[t]
                return dict(h=[tuple(bump(y) for y in x) if isinstance(x, tuple) else bump(x) for x in s['h']],
                            k=bump(s['k']), w=bump(s['w']))
            c2, av2 = R.cost_and_grads_checkpointed(p, cfg, feat, fm, lab, lm, spk, chunk=chunk, pinned=pinned)
            close = all(torch.allclose(p[k].grad, g, rtol=1e-9, atol=1e-13) for k, g in ref.items())
            assert close == same, (kw, scale)
            if same:
                assert abs(float(c2) - float(c)) < 1e-12 * abs(float(c))


def test_sample_step_equals_train_step_under_teacher_forcing():
    """sample_step fed with the data equals the training step."""
    cfg, p = _tiny(num_layers=2)
    from tests.util import make_batch
    T, B, U = 6, 2, 4
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=2)
    _, _, av, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, None, 1)
    outs = R.sample_model(p, cfg, lab, lm, None, T)
    # without feedback the decoder ignores x, so the two paths must agree exactly
    assert torch.allclose(outs[0], av[0], atol=1e-12)
    assert torch.allclose(outs[4], av[4], atol=1e-12)


def test_gru_step_matches_samplernn_twin():
    """Blocks GatedRecurrent restatement vs the in-repo twin algebra ops.py:364-393."""
    g = torch.Generator().manual_seed(0)
    H, B = 7, 3
    h = torch.randn(B, H, generator=g, dtype=torch.float64)
    x = torch.randn(B, 3 * H, generator=g, dtype=torch.float64)
    Wg = torch.randn(H, 2 * H, generator=g, dtype=torch.float64)
    Wc = torch.randn(H, H, generator=g, dtype=torch.float64)
    gates = torch.sigmoid(h @ Wg + x[:, :2 * H])
    update, reset = gates[:, :H], gates[:, H:]
    cand = torch.tanh((reset * h) @ Wc + x[:, 2 * H:])
    twin = update * cand + (1 - update) * h
    assert torch.allclose(R.gru_step(x[:, 2 * H:], x[:, :2 * H], h, Wc, Wg), twin, atol=1e-14)


def test_param_shapes_count_matches_survey():
    """W_step = 11.04 M for cfg2 (L=2) and 21.26 M for the 3-layer reference model."""
    for L, expect in ((2, 11.04e6), (3, 21.26e6)):
        cfg = R.default_config(num_layers=L, encoder_type='bidirectional')
        H, E, A = 1024, 256, 10
        n = sum((H + E + l * H) * 3 * H for l in range(L)) + 3 * A * H
        assert abs(n - expect) / expect < 0.01


def test_parrot_oracle_reproduces_frozen_vectors():
    """tests/golden/parrot_golden.npz was written by tests/golden/make_parrot_golden.py from this oracle: any
    later edit of oracle/parrot_ref.py that changes its arithmetic shows up here (GRU, GMM, LSTM, layer_norm)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mk", os.path.join(here, "golden", "make_parrot_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = np.load(os.path.join(here, "golden", "parrot_golden.npz"))
    for name, kw in mk.CASES.items():
        out = mk.run_case(kw)
        for k, v in out.items():
            np.testing.assert_allclose(v, gold[f"{name}|{k}"], rtol=1e-10, atol=1e-12, err_msg=f"{name}|{k}")
