"""The oracle restatements against REFERENCE-EXECUTED vectors (tests/golden/ref_golden.npz).

The fixture is produced by tests/golden/make_ref_golden.py, which runs the reference's own model.py / ops.py /
three_tier.py (unmodified, float64) on eager stand-ins for Theano and Blocks.  These tests are what pins
oracle/parrot_ref.py and oracle/samplernn_ref.py: every output and every parameter gradient must agree to 1e-10.
They need no /root/reference at run time (only the committed .npz)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.util import rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-10


def _mk():
    spec = importlib.util.spec_from_file_location("mk_ref", os.path.join(HERE, "golden", "make_ref_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_golden.npz"))


@pytest.fixture(scope="module")
def mk():
    return _mk()


def _close(a, b, what, tol=TOL):
    e = rel_err(torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b)))
    assert e <= tol, f"{what}: {e:.3e}"


def test_ops_level_vectors(gold, mk):
    """lib.ops.Linear (weight norm, two inputs), __GRUStep, __LSTMStep, Embedding, softmax_and_argmax as executed by the
    reference's own ops.py -> the oracle's linear / gru_step / lstm_step / argmax."""
    from oracle import samplernn_ref as S
    p = {k[len('ops|param:'):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('ops|param:')}
    x = {k[len('ops|in:'):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('ops|in:')}
    c = dict(S.DEFAULTS, WEIGHT_NORM=True)
    lin = S.linear(p, c, 'T.Lin', [x['x1'], x['x2']], n_inputs=2)
    _close(lin, gold['ops|out:linear'], 'Linear')
    _close(S.gru_step(p, c, 'T.GRU', 12, lin, x['h']), gold['ops|out:gru'], '__GRUStep')
    _close(S.lstm_step(p, c, 'T.LSTM', 12, lin, x['hc']), gold['ops|out:lstm'], '__LSTMStep')
    _close(p['T.Emb'][x['idx']], gold['ops|out:embedding'], 'Embedding')
    am = torch.argmax(torch.softmax(x['logits'], -1), -1)
    assert np.array_equal(am.numpy(), gold['ops|out:argmax'])
    assert gold['ops|out:argmax'][0, 0] == 2  # the planted tie resolves to the lowest index


def test_blocks_shim_gru_equals_reference_grustep(gold):
    """The Blocks GatedRecurrent algebra restated in oracle/refshim/blocks_shim.py (and in oracle/parrot_ref.gru_step)
    equals the reference's in-repo twin __GRUStep (ops.py:364-393) on the reference-executed vector: same update-first
    gate order, same z*c + (1-z)*h blend."""
    from oracle import parrot_ref as R
    from oracle import samplernn_ref as S
    p = {k[len('ops|param:'):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('ops|param:')}
    x = {k[len('ops|in:'):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('ops|in:')}
    c = dict(S.DEFAULTS, WEIGHT_NORM=True)
    D = 12
    lin = S.linear(p, c, 'T.Lin', [x['x1'], x['x2']], n_inputs=2)
    eff = lambda n: p[n + '.W0'] * (p[n + '.g0'] / p[n + '.W0'].norm(dim=0))[None, :]  # noqa: E731
    pi = lin @ eff('T.GRU.Input') + p['T.GRU.Input.b']
    h = R.gru_step(pi[:, 2 * D:], pi[:, :2 * D], x['h'], eff('T.GRU.Recurrent_Candidate'), eff('T.GRU.Recurrent_Gates'))
    _close(h, gold['ops|out:gru'], 'Blocks GatedRecurrent vs __GRUStep')


@pytest.mark.parametrize("case", ["gru1", "lstm2", "gru2", "gru2skip", "lstm3skip"])
def test_three_tier_compute_cost_and_gradients(gold, mk, case):
    from oracle import samplernn_ref as S
    skip = case in mk.SR_SKIP_CASES  # stacks with skip connections (ops.py:650-695, 861-880)
    rnn, n = (mk.SR_SKIP_CASES if skip else mk.SR_CASES)[case]
    c, p = mk.sr_params(rnn, n, skip)
    seq, feats, h0, bh0, mask = mk.sr_inputs(rnn, n)
    for reset in (0, 1):
        pre = f'sr:{case}:r{reset}|'
        rp = {k: v.clone().requires_grad_() for k, v in p.items()}
        cost, ip, nh0, nbh0 = S.compute_cost(rp, c, seq, feats, h0, bh0, reset, mask)
        (cost + ip).backward()
        _close(cost.detach(), gold[pre + 'cost'], 'cost')
        _close(ip.detach(), gold[pre + 'ip_cost'], 'ip_cost')
        _close(nh0.detach(), gold[pre + 'new_h0'], 'new_h0')
        _close(nbh0.detach(), gold[pre + 'new_big_h0'], 'new_big_h0')
        n_grads = 0
        for k in p:
            key = pre + 'grad:' + k
            if key not in gold.files:
                assert rp[k].grad is None or float(rp[k].grad.abs().max()) == 0.0, k
                continue
            _close(mk.pack_grad(k, rp[k].grad), gold[key], f'grad {k}', 1e-9)
            n_grads += 1
        assert n_grads >= 40


def test_three_tier_generation_loop(gold, mk):
    """The reference's generate_and_save_samples loop (temperature-0 sampler) vs oracle.generate."""
    from oracle import samplernn_ref as S
    c, p = mk.sr_params('GRU', 1)
    with torch.no_grad():
        out = S.generate(p, c, mk.gen_features()).numpy()
    ref = gold['sr:gru1|samples']
    assert ref.shape == (3, 320) and (ref[:, :80] == 128).all()
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("case", ["base", "fb_spk", "softmax_ln", "gmm", "sharp"])
def test_parrot_compute_cost_gradients_carry_and_decode(gold, mk, case):
    from oracle import parrot_ref as R
    full, cfg, p, (feat, fm, lab, lm, spk) = mk.par_setup(mk.PAR_CASES[case])
    pre = f'par:{case}|'
    rp = {k: v.clone().requires_grad_() for k, v in p.items()}
    cost, carry, av, _ = R.compute_cost(rp, cfg, feat, fm, lab, lm, spk, 1)
    cost.backward()
    _close(cost.detach(), gold[pre + 'cost'], 'cost')
    for i, n in enumerate(('next_x', 'k', 'w', 'coeff', 'phi', 'pi_att')):
        if pre + n in gold.files:
            _close(av[i].detach(), gold[pre + n], n)
    for l in range(3):
        _close(carry['h'][l].detach(), gold[pre + f'carry:h{l + 1}'], f'carry h{l + 1}')
    _close(carry['k'].detach(), gold[pre + 'carry:k'], 'carry k')
    _close(carry['w'].detach(), gold[pre + 'carry:w'], 'carry w')
    for k in p:
        g = rp[k].grad if rp[k].grad is not None else torch.zeros_like(p[k])
        _close(mk.pack_grad(k, g), gold[pre + 'grad:' + k], f'grad {k}', 1e-9)
    with torch.no_grad():
        c1, carry1, _, _ = R.compute_cost(p, cfg, feat[:5], fm[:5], lab, lm, spk, 1)
        c2, _, av2, _ = R.compute_cost(p, cfg, feat[4:], fm[4:], lab, lm, spk, 0, carry=carry1)
    _close(c1, gold[pre + 'w1:cost'], 'window-1 cost')
    _close(c2, gold[pre + 'w2:cost'], 'window-2 cost (start_flag = 0, carried state)')
    _close(av2[1], gold[pre + 'w2:k'], 'window-2 kappa')
    if cfg['which_cost'] == 'MSE':
        _close(av2[0], gold[pre + 'w2:next_x'], 'window-2 frames')
        with torch.no_grad():
            sm = R.sample_model(p, cfg, lab, lm, spk, mk.PAR_S)
        for o, n in zip(sm, ('sample_x', 'k', 'w', 'pi', 'phi', 'pi_att')):
            _close(o, gold[pre + 'sample:' + n], 'sample ' + n)
