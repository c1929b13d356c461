"""The HIP path against REFERENCE-EXECUTED vectors (tests/golden/ref_golden.npz, produced by running the reference's own
model.py / ops.py / three_tier.py on eager Theano / Blocks stand-ins, see tests/golden/make_ref_golden.py).
Tolerances: 1e-4 relative on outputs (north star), 2e-3 norm-wise per gradient (fp32 accumulation vs a float64
reference), integer outputs bit-exact."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mk():
    spec = importlib.util.spec_from_file_location("mk_ref", os.path.join(HERE, "golden", "make_ref_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_golden.npz"))


def _g(gold, key):
    return torch.from_numpy(np.asarray(gold[key]))


def test_ops_level_hip_vs_reference_ops(dev, gold):
    """HIP Linear (weight norm, two inputs), one GRU step and one LSTM step (LowMemGRU / LowMemLSTM on a length-1
    sequence) vs the reference's own ops.py outputs."""
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.lib import ops as lops
    lib.delete_all_params()
    lib.set_device(dev)
    try:
        lib.set_params({k[len('ops|param:'):]: _g(gold, k) for k in gold.files if k.startswith('ops|param:')})
        x = {k[len('ops|in:'):]: _g(gold, k) for k in gold.files if k.startswith('ops|in:')}
        f = lambda t: t.float().to(dev)  # noqa: E731
        lin = lops.Linear('T.Lin', [7, 3], 12, [f(x['x1']), f(x['x2'])], initialization='he', weightnorm=True)
        assert_close(lin, _g(gold, 'ops|out:linear'), 1e-5, "Linear")
        ref_lin = f(_g(gold, 'ops|out:linear'))
        # reference-name aliases: the scan wrappers register `<name>.Step.*`
        for k in list(lib.named_params()):
            if k.startswith('T.GRU.') or k.startswith('T.LSTM.'):
                head, rest = k.split('.', 2)[1], k.split('.', 2)[2]
                lib.set_params({f'T.{head}.Step.{rest}': lib.param(k).detach().cpu().double()})
        gru = lops.LowMemGRU('T.GRU', 12, 12, ref_lin[:, None, :], h0=f(x['h']), weightnorm=True)[:, 0]
        assert_close(gru, _g(gold, 'ops|out:gru'), 1e-5, "__GRUStep")
        lstm = lops.LowMemLSTM('T.LSTM', 12, 12, ref_lin[:, None, :], h0=f(x['hc']), weightnorm=True)[:, 0]
        assert_close(lstm, _g(gold, 'ops|out:lstm'), 1e-5, "__LSTMStep")
        emb = lops.Embedding('T.Emb', 9, 5, x['idx'].to(dev))
        assert_close(emb, _g(gold, 'ops|out:embedding'), 1e-6, "Embedding")
        am = lops.softmax_and_argmax(f(x['logits']))
        assert np.array_equal(am.cpu().numpy(), gold['ops|out:argmax'])
    finally:
        lib.delete_all_params()


@pytest.mark.parametrize("case", ["gru1", "lstm2", "gru2", "gru2skip", "lstm3skip"])
def test_three_tier_hip_vs_reference(dev, gold, mk, case):
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    skip = case in mk.SR_SKIP_CASES  # stacks with skip connections (ops.py:650-695, 861-880)
    rnn, n = (mk.SR_SKIP_CASES if skip else mk.SR_CASES)[case]
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=mk.SR_DIM, EMB_SIZE=mk.SR_EMB, RNN_TYPE=rnn, N_RNN=n, SKIP_CONN=skip)
    try:
        c, p = mk.sr_params(rnn, n, skip)
        lib.set_params(p)
        seq, feats, h0, bh0, mask = mk.sr_inputs(rnn, n)
        for reset in (0, 1):
            pre = f'sr:{case}:r{reset}|'
            for t in lib.named_params().values():
                t.grad = None
            cost, ip_cost, allp, ipp, otherp, nh0, nbh0 = tt.compute_cost(
                seq.to(dev), feats.float().to(dev), h0.float().to(dev), bh0.float().to(dev), reset, mask.float().to(dev))
            (cost + ip_cost).backward()
            assert_close(cost, _g(gold, pre + 'cost'), 1e-4, "cost")
            assert_close(ip_cost, _g(gold, pre + 'ip_cost'), 1e-4, "ip_cost")
            assert_close(nh0, _g(gold, pre + 'new_h0'), 1e-4, "new_h0")
            assert_close(nbh0, _g(gold, pre + 'new_big_h0'), 1e-4, "new_big_h0")
            assert [len(allp), len(ipp), len(otherp)] == gold[pre + 'n_params'].tolist()
            checked = 0
            for name, t in lib.named_params().items():
                key = pre + 'grad:' + name
                if key not in gold.files or float(np.abs(gold[key]).max()) < 1e-12:
                    continue
                assert t.grad is not None, name
                assert_close(mk.pack_grad(name, t.grad.detach().cpu()), _g(gold, key), 2e-3, f"grad {name}")
                checked += 1
            assert checked >= 40
        if case == 'gru1':
            gen = tt.DeviceGenerator(3, 4, temperature=0.0, use_graph=True)
            out = gen.generate(mk.gen_features().float().numpy()).cpu().numpy()
            gen.close()
            assert np.array_equal(out, gold['sr:gru1|samples']), "greedy indices vs the reference's own sample loop"
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1, SKIP_CONN=False)


@pytest.mark.parametrize("case", ["base", "fb_spk", "softmax_ln", "gmm", "sharp"])
def test_parrot_hip_vs_reference_model_py(dev, gold, mk, case):
    """Parrot.compute_cost (cost, frames, window state, every gradient, TBPTT carry) and sample_model at the reference's
    own depth (3 GRU layers) vs the reference's model.py executed on the shims."""
    from parrot_amd.model import Parrot
    full, cfg, p, (feat, fm, lab, lm, spk) = mk.par_setup(mk.PAR_CASES[case])
    pre = f'par:{case}|'
    m = Parrot(device=dev, use_graph=True, num_layers=3, **full).allocate()
    m.set_parameter_values(p)
    sp = None if spk is None else spk.to(dev)
    args = (lab.to(dev), lm.float().to(dev), sp)
    for rep in range(2):
        m.zero_grad()
        cost, upd, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), *args, 1, mk.PAR_B)
        cost.backward()
        assert_close(cost, _g(gold, pre + 'cost'), 1e-4, "cost")
        for i, n in enumerate(('next_x', 'k', 'w', 'coeff', 'phi', 'pi_att')):
            if pre + n in gold.files and n != 'coeff':
                assert_close(av[i], _g(gold, pre + n), 1e-4, n)
        grads = m.get_gradient_dict()
        tol = 2e-3 if (cfg['layer_norm'] or cfg['which_cost'] == 'GMM') else 1e-3
        for k in p:
            ref = _g(gold, pre + 'grad:' + k)
            if float(ref.abs().max()) < 1e-12:
                continue
            assert_close(mk.pack_grad(k, grads[k].detach().cpu()), ref, tol, f"grad {k}")
    c1, u1, _, _ = m.compute_cost(feat[:5].float().to(dev), fm[:5].float().to(dev), *args, 1, mk.PAR_B)
    assert_close(c1, _g(gold, pre + 'w1:cost'), 1e-4, "window-1 cost")
    m.apply_updates(u1)
    c2, _, av2, _ = m.compute_cost(feat[4:].float().to(dev), fm[4:].float().to(dev), *args, 0, mk.PAR_B)
    assert_close(c2, _g(gold, pre + 'w2:cost'), 1e-4, "window-2 cost (carried state)")
    assert_close(av2[1], _g(gold, pre + 'w2:k'), 1e-4, "window-2 kappa")
    if cfg['which_cost'] == 'MSE':
        assert_close(av2[0], _g(gold, pre + 'w2:next_x'), 1e-4, "window-2 frames")
        outs = m.sample_model_device(lab, lm.float(), spk, mk.PAR_B, mk.PAR_S)
        for o, n in zip(outs, ('sample_x', 'k', 'w', 'pi', 'phi', 'pi_att')):
            assert_close(o, _g(gold, pre + 'sample:' + n), 1e-4, "sample " + n)
    m.close()
