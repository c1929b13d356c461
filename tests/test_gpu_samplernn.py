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


def _greedy_follows_oracle(out, ref, ref_logits, B, slack_rows):
    """Rows equal the oracle's trajectory; a row may leave it only where the oracle's own top-2 logits are closer than
    fp32 resolves (gap < 2e-5 of the logit scale).  Returns the rows that are identical."""
    exact = []
    for b in range(B):
        diff = np.nonzero(out[b] != ref[b])[0]
        if diff.size == 0:
            exact.append(b)
            continue
        t = int(diff[0])
        lg = ref_logits[b, t - 80]
        top2 = torch.topk(lg, 2).values
        gap = float(top2[0] - top2[1])
        assert gap < 2e-5 * float(lg.abs().max()), f"row {b} leaves the oracle at sample {t}: top-2 gap {gap:.3e}"
    assert len(exact) >= B - slack_rows, f"only {len(exact)} of {B} rows follow the oracle"
    return exact


@pytest.mark.parametrize("dim,B,T", [(256, 3, 4), (256, 32, 3), (512, 17, 3), (256, 1, 3)])
def test_persistent_sample_kernel_matches_oracle_and_launch_path(dev, dim, B, T, monkeypatch):
    """sr_persist.hip (one launch per 10 sample steps, XCD-local teams) vs the fp64 oracle and vs the five-launches-per-
    sample path (three_tier.py:452-515, 809-832): greedy indices, graph replay and eager launches, partial teams
    (B not a multiple of 4), two calls on one plan and a second utterance on it (what a launch leaves for the next one --
    slot states, the carried history and gather -- is either valid or recognised as stale)."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=dim, EMB_SIZE=32, RNN_TYPE='GRU', N_RNN=1)
    try:
        c = S.config(DIM=dim, EMB_SIZE=32)
        p = S.init_params(c, seed=11, perturb=0.25)
        lib.set_params(p)
        g = torch.Generator().manual_seed(6)
        feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref, ref_logits = S.generate(p, c, feats, return_logits=True)
        ref = ref.numpy()
        outs = {}
        for use_graph in (True, False):
            gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=use_graph)
            assert gen.persistent, "the persistent sample kernel did not engage on a qualifying shape"
            for rep in range(2):
                out = gen.generate(feats.float().numpy()).cpu().numpy()
                exact = _greedy_follows_oracle(out, ref, ref_logits, B, slack_rows=1)
                last = gen.ws['logits'].detach().cpu().double()
                assert_close(last[exact], ref_logits[exact, -1], 1e-4, "last-step logits")
            outs[use_graph] = out
            # another utterance on the same plan: the carry the last launch left behind (history + first gather of a
            # launch that never comes) must not leak into it -- same samples as a fresh plan produces
            feats2 = torch.randn(T, B, 63, generator=torch.Generator().manual_seed(7)).numpy()
            again = gen.generate(feats2).cpu().numpy().copy()
            gen.close()
            fresh = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=use_graph)
            assert np.array_equal(again, fresh.generate(feats2).cpu().numpy())
            fresh.close()
        assert np.array_equal(outs[True], outs[False])
        monkeypatch.setenv("PARROT_SR_PERSIST", "0")
        gen = tt.DeviceGenerator(B, T, temperature=0.0)
        assert not gen.persistent
        launch = gen.generate(feats.float().numpy()).cpu().numpy()
        gen.close()
        same = [b for b in range(B) if np.array_equal(launch[b], ref[b]) and np.array_equal(outs[True][b], ref[b])]
        assert len(same) >= B - 1
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)


@pytest.mark.parametrize("rnn_type,n_rnn", [("GRU", 2), ("LSTM", 1), ("LSTM", 2)])
def test_persistent_sample_kernel_under_stacked_and_lstm_tiers(dev, rnn_type, n_rnn):
    """Stacked / LSTM tiers keep their frame tier on launches, but the sample steps still run on the persistent kernel and
    so on the COMPOSED projection (frame_out = top . (Wout . W2) + bout . W2 + b2, sr_persist.hip): greedy indices and the
    last step's logits vs the fp64 oracle at a width the kernel takes (DIM 256), partial team (B = 5)."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=256, EMB_SIZE=32, RNN_TYPE=rnn_type, N_RNN=n_rnn)
    try:
        c = S.config(DIM=256, EMB_SIZE=32, RNN_TYPE=rnn_type, N_RNN=n_rnn)
        p = S.init_params(c, seed=13, perturb=0.25)
        lib.set_params(p)
        g = torch.Generator().manual_seed(8)
        T, B = 3, 5
        feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref, ref_logits = S.generate(p, c, feats, return_logits=True)
        ref = ref.numpy()
        gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=True)
        assert gen.persistent, "the persistent sample kernel did not engage"
        for rep in range(2):
            out = gen.generate(feats.float().numpy()).cpu().numpy()
            exact = _greedy_follows_oracle(out, ref, ref_logits, B, slack_rows=1)
            last = gen.ws['logits'].detach().cpu().double()
            assert_close(last[exact], ref_logits[exact, -1], 1e-4, "last-step logits")
        gen.close()
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)


def test_persistent_sample_kernel_seeded_draws(dev):
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=256, EMB_SIZE=32, RNN_TYPE='GRU', N_RNN=1)
    try:
        c = S.config(DIM=256, EMB_SIZE=32)
        lib.set_params(S.init_params(c, seed=12, perturb=0.25))
        feats = np.random.RandomState(1).randn(3, 6, 63).astype('float32')
        outs = []
        for seed in (77, 77, 78):
            gen = tt.DeviceGenerator(6, 3, temperature=1.0, seed=seed)
            assert gen.persistent
            outs.append(gen.generate(feats).cpu().numpy().copy())
            gen.close()
        assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
        assert outs[0].min() >= 0 and outs[0].max() <= 255
        assert len(np.unique(outs[0][:, 80:])) > 20
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)


@pytest.mark.parametrize("rnn_type,n_rnn", [("GRU", 2), ("LSTM", 2)])
def test_generation_with_skip_connection_stacks(dev, rnn_type, n_rnn):
    """--skip_conn True (ops.py:650-695, 861-880): the tiers' stacks feed the stack input to every layer and sum per-layer
    output projections.  Generation runs on the literal three-function loop (three_tier.py:809-832); greedy indices equal
    the fp64 oracle's."""
    from oracle import samplernn_ref as S
    from parrot_amd.sampleRNN import lib
    from parrot_amd.sampleRNN.models.conditional import three_tier as tt
    lib.delete_all_params()
    lib.set_device(dev)
    tt.configure(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn, SKIP_CONN=True)
    try:
        c = S.config(DIM=32, EMB_SIZE=8, RNN_TYPE=rnn_type, N_RNN=n_rnn, SKIP_CONN=True)
        p = S.init_params(c, seed=9, perturb=0.3)
        lib.set_params(p)
        g = torch.Generator().manual_seed(4)
        T, B = 3, 2
        feats = torch.randn(T, B, 63, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref = S.generate(p, c, feats).numpy()
        fns = tt.getting_generation_functions()
        out = tt.generate_and_save_samples("t", None, feats.float().numpy(), None, 0., *fns, temperature=0.0)
        assert np.array_equal(out, ref), f"{(out != ref).sum()} of {out.size} greedy indices differ"
        with pytest.raises(NotImplementedError):
            tt.DeviceGenerator(B, T)
    finally:
        lib.delete_all_params()
        tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1, SKIP_CONN=False)
