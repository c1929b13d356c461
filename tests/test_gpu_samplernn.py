"""Parity of the conditional three-tier SampleRNN (HIP path) against the fp64 oracle restatement of
sampleRNN/lib/ops.py + three_tier.py: training cost / gradients and the greedy generation loop."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def small_model(dev):
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=64, EMB_SIZE=16)
    c = S.config(DIM=64, EMB_SIZE=16)
    p = S.init_params(c, seed=5, perturb=0.2)
    lib.set_params(p)
    yield c, p, tt, lib
    lib.delete_all_params()
    tt.configure(DIM=1024, EMB_SIZE=256)


def test_compute_cost_and_grads(dev, small_model):
    from oracle import samplernn_ref as S
    c, p, tt, lib = small_model
    g = torch.Generator().manual_seed(1)
    B, S_len = 3, 160
    seq = torch.randint(0, 256, (B, S_len + 80), generator=g)
    feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
    mask = torch.ones(B, S_len + 80, dtype=torch.float64)
    mask[1, 200:] = 0
    for reset in (1, 0):
        h0 = torch.randn(B, 1, 64, generator=g, dtype=torch.float64) * 0.3
        bh0 = torch.randn(B, 1, 64, generator=g, dtype=torch.float64) * 0.3
        ref_p = {k: v.clone().requires_grad_() for k, v in p.items()}
        rc, rip, rh0, rbh0 = S.compute_cost(ref_p, c, seq, feats, h0, bh0, reset, mask)
        (rc + rip).backward()
        for t in lib.named_params().values():
            t.grad = None
        cost, ip_cost, allp, ipp, otherp, nh0, nbh0 = tt.compute_cost(
            seq.to(dev), feats.float().to(dev), h0.float().to(dev), bh0.float().to(dev), reset, mask.float().to(dev))
        (cost + ip_cost).backward()
        assert_close(cost, rc, 1e-4, "cost")
        assert_close(ip_cost, rip, 1e-4, "ip_cost")
        assert_close(nh0, rh0, 1e-4, "new_h0")
        assert_close(nbh0, rbh0, 1e-4, "new_big_h0")
        for name, t in lib.named_params().items():
            rg = ref_p[name].grad
            if rg is None or float(rg.abs().max()) < 1e-12:
                continue
            assert t.grad is not None, name
            assert rel_err(t.grad, rg) < 2e-3, (name, rel_err(t.grad, rg))
    assert len(allp) == len(ipp) + len(otherp)


def test_device_generation_matches_oracle_and_python_loop(dev, small_model):
    from oracle import samplernn_ref as S
    c, p, tt, lib = small_model
    g = torch.Generator().manual_seed(2)
    T, B = 4, 3
    feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
    with torch.no_grad():
        ref = S.generate(p, c, feats).numpy()
    for use_graph in (True, False):
        gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=use_graph)
        out = gen.generate(feats.float().numpy()).cpu().numpy()
        gen.close()
        assert out.shape == ref.shape and out.dtype == np.int32
        assert (out[:, :80] == 128).all()
        assert np.array_equal(out, ref), f"{(out != ref).sum()} of {out.size} greedy indices differ"
    fns = tt.getting_generation_functions()
    lit = tt.generate_and_save_samples("t", None, feats.float().numpy(), None, 0., *fns, temperature=0.0,
                                       use_device_loop=False)
    assert np.array_equal(lit, ref)


def test_stochastic_sampling_is_seeded_and_in_range(dev, small_model):
    c, p, tt, lib = small_model
    feats = np.random.RandomState(0).randn(3, 2, 63).astype('float32')
    outs = []
    for _ in range(2):
        gen = tt.DeviceGenerator(2, 3, temperature=1.0, seed=77)
        outs.append(gen.generate(feats).cpu().numpy().copy())
        gen.close()
    assert np.array_equal(outs[0], outs[1])
    assert outs[0].min() >= 0 and outs[0].max() <= 255
    assert len(np.unique(outs[0][:, 80:])) > 20  # it really samples


def test_param_registry_roundtrip(dev, small_model, tmp_path):
    c, p, tt, lib = small_model
    path = str(tmp_path / "params.pkl")
    lib.save_params(path)
    w = lib.param('SampleLevel.L2.W0')
    before = w.detach().clone()
    with torch.no_grad():
        w.zero_()
    lib.load_params(path)
    assert torch.equal(lib.param('SampleLevel.L2.W0'), before)
    assert lib.param('SampleLevel.L2.W0') is w  # shared object between graphs (lib/__init__.py:28-47)


@pytest.mark.parametrize("rnn_type,n_rnn", [("LSTM", 1), ("LSTM", 2), ("GRU", 2)])
def test_compute_cost_lstm_and_stacked(dev, rnn_type, n_rnn):
    """RNN_TYPE = 'LSTM' (ops.py:461-610, 823-989) and stacked (n_rnn > 1) variants of the tiers."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn)
    try:
        c = S.config(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn)
        p = S.init_params(c, seed=9, perturb=0.2)
        lib.set_params(p)
        g = torch.Generator().manual_seed(3)
        B, S_len, hm = 2, 160, c['H0_MULT']
        seq = torch.randint(0, 256, (B, S_len + 80), generator=g)
        feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
        mask = torch.ones(B, S_len + 80, dtype=torch.float64)
        h0 = torch.randn(B, n_rnn, hm * 32, generator=g, dtype=torch.float64) * 0.3
        bh0 = torch.randn(B, n_rnn, hm * 32, generator=g, dtype=torch.float64) * 0.3
        ref_p = {k: v.clone().requires_grad_() for k, v in p.items()}
        rc, rip, rh0, rbh0 = S.compute_cost(ref_p, c, seq, feats, h0, bh0, 0, mask)
        (rc + rip).backward()
        cost, ip_cost, _, _, _, nh0, nbh0 = tt.compute_cost(seq.to(dev), feats.float().to(dev), h0.float().to(dev),
                                                            bh0.float().to(dev), 0, mask.float().to(dev))
        (cost + ip_cost).backward()
        assert_close(cost, rc, 1e-4, "cost")
        assert_close(nh0, rh0, 1e-4, "new_h0")
        assert_close(nbh0, rbh0, 1e-4, "new_big_h0")
        assert set(lib.named_params()) == set(p), set(lib.named_params()) ^ set(p)
        for name, t in lib.named_params().items():
            rg = ref_p[name].grad
            if rg is None or float(rg.abs().max()) < 1e-12:
                continue
            assert t.grad is not None and rel_err(t.grad, rg) < 2e-3, (name, rel_err(t.grad, rg))
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)


@pytest.mark.parametrize("rnn_type,n_rnn", [("LSTM", 1), ("LSTM", 2), ("GRU", 2), ("GRU", 3)])
def test_device_generation_lstm_and_stacked_tiers(dev, rnn_type, n_rnn):
    """The device-resident sample loop for RNN_TYPE = 'LSTM' and stacked tiers (three_tier.py:147-169; stackedGRU /
    stackedLSTM, ops.py:612-777, 823-989): greedy indices equal the fp64 oracle's, graph replay and eager launches."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn)
    try:
        c = S.config(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn)
        p = S.init_params(c, seed=9, perturb=0.3)
        lib.set_params(p)
        g = torch.Generator().manual_seed(4)
        T, B = 4, 3
        feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref = S.generate(p, c, feats).numpy()
        for use_graph in (True, False):
            gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=use_graph)
            for rep in range(2):  # the second call starts again from the learned h0
                out = gen.generate(feats.float().numpy()).cpu().numpy()
                assert np.array_equal(out, ref), f"{(out != ref).sum()} of {out.size} greedy indices differ"
            gen.close()
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
