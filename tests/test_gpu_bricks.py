"""Every brick of parrot_amd/bricks.py (the Blocks surface SURVEY 8b lists) called stand-alone through its `apply`,
forward AND backward, against the oracle (oracle/parrot_ref.py, pinned to reference-executed vectors)."""
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _grads(outs, leaves, seed=0):
    g = torch.Generator().manual_seed(seed)
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    loss = 0
    for o in outs:
        w = torch.randn(o.shape, generator=g, dtype=torch.float64).to(o.device, o.dtype)
        loss = loss + (o * w).sum()
    return torch.autograd.grad(loss, leaves, allow_unused=True)


def test_linear_fork_lookup_apply_and_gradients(dev):
    from oracle import parrot_ref as R
    from parrot_amd.bricks import Fork, Linear, LookupTable
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 5, 20, generator=g, dtype=torch.float64)
    p = {'/parrot/lin.W': torch.randn(20, 33, generator=g, dtype=torch.float64) * 0.3,
         '/parrot/lin.b': torch.randn(33, generator=g, dtype=torch.float64),
         '/parrot/fk/fork_a.W': torch.randn(20, 16, generator=g, dtype=torch.float64) * 0.3,
         '/parrot/fk/fork_a.b': torch.randn(16, generator=g, dtype=torch.float64),
         '/parrot/fk/fork_b.W': torch.randn(20, 32, generator=g, dtype=torch.float64) * 0.3,
         '/parrot/fk/fork_b.b': torch.randn(32, generator=g, dtype=torch.float64)}
    for v in p.values():
        v.requires_grad_()
    xr = x.clone().requires_grad_()
    ref = [R.linear(p, 'lin', xr)] + R.fork(p, 'fk', xr, ['a', 'b'])
    rg = _grads(ref, [xr] + list(p.values()))
    lin = Linear(20, 33, name='lin', device=dev)
    fk = Fork(['a', 'b'], 20, [16, 32], name='fk', device=dev)
    named = dict(lin.get_parameter_dict('/parrot'))
    named.update(fk.get_parameter_dict('/parrot'))
    assert set(named) == set(p)
    with torch.no_grad():
        for k, v in p.items():
            named[k].copy_(v.float())
    xd = x.float().to(dev).requires_grad_()
    out = [lin.apply(xd)] + fk.apply(xd)
    assert list(fk.apply(xd, as_dict=True).keys()) == ['a', 'b']
    for o, r, n in zip(out, ref, ('linear', 'fork a', 'fork b')):
        assert_close(o, r, 1e-5, n)
    hg = _grads(out, [xd] + [named[k] for k in p])
    for a, b, n in zip(hg, rg, ['x'] + list(p)):
        assert_close(a, b, 1e-4, 'grad ' + n)
    tab = LookupTable(9, 6, device=dev)
    with torch.no_grad():
        tab.W.copy_(torch.randn(9, 6, generator=g))
    idx = torch.randint(0, 9, (4, 3), generator=g)
    assert torch.equal(tab.apply(idx.to(dev)).cpu(), tab.W.detach().cpu()[idx])
    assert list(tab.get_parameter_dict('/parrot')) == ['/parrot/lookuptable.W']


@pytest.mark.parametrize("masked", [False, True])
def test_gated_recurrent_step_and_scan(dev, masked):
    from oracle import parrot_ref as R
    from parrot_amd.bricks import GatedRecurrent
    g = torch.Generator().manual_seed(2)
    T, B, D = 6, 5, 48
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    Wss, Wsg, h0 = (rn(D, D) * 0.2).requires_grad_(), (rn(D, 2 * D) * 0.2).requires_grad_(), (rn(D) * 0.5).requires_grad_()
    inp, gat = rn(T, B, D).requires_grad_(), rn(T, B, 2 * D).requires_grad_()
    mask = (torch.rand(T, B, generator=g) > 0.3).double() if masked else None
    ref_seq = R.gru_scan(inp, gat, h0.expand(B, -1), Wss, Wsg, mask)
    ref_step = R.gru_step(inp[0], gat[0], h0.expand(B, -1), Wss, Wsg, None if mask is None else mask[0])
    rg = _grads([ref_seq, ref_step], [inp, gat, Wss, Wsg, h0])
    gru = GatedRecurrent(D, name='rnn', device=dev)
    names = list(gru.get_parameter_dict('/parrot'))
    assert names == ['/parrot/rnn.state_to_state', '/parrot/rnn.state_to_gates', '/parrot/rnn.initial_state']
    with torch.no_grad():
        gru.parameters['state_to_state'].copy_(Wss.float())
        gru.parameters['state_to_gates'].copy_(Wsg.float())
        gru.parameters['initial_state'].copy_(h0.float())
    f = lambda t: t.detach().float().to(dev).requires_grad_()  # noqa: E731
    inp_d, gat_d = f(inp), f(gat)
    md = None if mask is None else mask.float().to(dev)
    seq = gru.apply(inp_d, gat_d, mask=md)  # iterate=True from initial_states
    step = gru.apply(inp_d[0], gat_d[0], gru.initial_states(B), None if md is None else md[0], iterate=False)
    assert_close(seq, ref_seq, 1e-5, "scan states")
    assert_close(step, ref_step, 1e-5, "single step")
    hg = _grads([seq, step], [inp_d, gat_d, gru.parameters['state_to_state'], gru.parameters['state_to_gates'],
                              gru.parameters['initial_state']])
    for a, b, n in zip(hg, rg, ('inputs', 'gate_inputs', 'state_to_state', 'state_to_gates', 'initial_state')):
        assert_close(a, b, 2e-4, 'grad ' + n)


def test_bidirectional_recurrent_with_fork_is_the_reference_encoder(dev):
    """Bidirectional(RecurrentWithFork(GatedRecurrent)) + LookupTable assembled from the bricks == Encoder.apply
    (model.py:201-247), literal mode (scan over axis 0 of the batch-major embedding), forward and backward."""
    from oracle import parrot_ref as R
    from parrot_amd.bricks import Bidirectional, GatedRecurrent, LookupTable, RecurrentWithFork
    cfg = R.default_config(encoder_type='bidirectional', encoder_dim=16, input_dim=12, rnn_h_dim=8, readouts_dim=8,
                           num_layers=1)
    p = {k: v for k, v in R.init_params(cfg, seed=4, scale_by_fan_in=True).items() if '/encoder/' in k}
    for v in p.values():
        v.requires_grad_()
    labels = torch.randint(0, 43, (5, 9), generator=torch.Generator().manual_seed(3))
    ref = R.encoder_apply(p, cfg, labels)
    rg = _grads(ref, list(p.values()))
    table = LookupTable(43, 12, name='embed_label', device=dev)
    enc = Bidirectional(RecurrentWithFork(GatedRecurrent(16, device=dev), 12, name='encoder_transition', device=dev),
                        name='encoder', device=dev)
    named = dict(table.get_parameter_dict('/parrot/encoder'))
    named.update(enc.get_parameter_dict('/parrot/encoder'))
    assert set(named) == set(p), set(named) ^ set(p)
    with torch.no_grad():
        for k, v in p.items():
            named[k].copy_(v.float())
    out = enc.apply(table.apply(labels.to(dev)))
    assert_close(out, ref, 1e-5, "encoder output")
    hg = _grads(out, [named[k] for k in p])
    for a, b, n in zip(hg, rg, p):
        if b is None or float(b.abs().max()) < 1e-12:
            continue
        assert_close(a, b, 2e-4, 'grad ' + n)
