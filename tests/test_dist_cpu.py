"""world_size-2 gloo tests of the data-parallel plumbing (CPU).  The gradient provider is the oracle:
what is under test is parrot_amd.dist (sharding, global-denominator scaling, flat all-reduce)."""
import os

import pytest
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from oracle import parrot_ref as R
    from parrot_amd import dist as pdist
    from tests.util import make_batch
    torch.set_num_threads(1)
    r, lr, w = pdist.init_process_group(backend="gloo")
    assert (r, w) == (rank, world) and pdist.is_distributed()
    cfg = R.default_config(rnn_h_dim=12, readouts_dim=10, encoder_type='bidirectional', encoder_dim=4, input_dim=6,
                           num_layers=2, encoder_literal=False)
    p = R.init_params(cfg, seed=3, scale_by_fan_in=True)
    names = sorted(p)
    T, B, U = 5, 6, 4
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=1, ragged=True)
    lo, hi = pdist.shard_batch(B, rank, world)
    for v in p.values():
        v.requires_grad_()
    cost, _, _, _ = R.compute_cost(p, cfg, feat[:, lo:hi], fm[:, lo:hi], lab[lo:hi], lm[lo:hi], None, 1)
    den_local = fm[1:, lo:hi].sum()
    scale, den_global = pdist.global_cost_scale(den_local)
    cost.backward(gradient=scale.to(cost.dtype))
    flat = torch.cat([p[n].grad.reshape(-1) for n in names]).float()
    pdist.allreduce_flat_(flat)
    gcost = pdist.allreduce_cost(cost.detach() * (den_local + pdist.COST_EPS), den_global)
    pdist.barrier()
    if rank == 0:
        ret["flat"] = flat.clone()
        ret["cost"] = float(gcost)
        ret["den"] = float(den_global)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_gradients_equal_single_process():
    from oracle import parrot_ref as R
    from tests.util import make_batch
    world, port = 2, 29517
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    cfg = R.default_config(rnn_h_dim=12, readouts_dim=10, encoder_type='bidirectional', encoder_dim=4, input_dim=6,
                           num_layers=2, encoder_literal=False)
    p = R.init_params(cfg, seed=3, scale_by_fan_in=True)
    names = sorted(p)
    feat, fm, lab, lm, _ = make_batch(cfg, 5, 6, 4, seed=1, ragged=True)
    for v in p.values():
        v.requires_grad_()
    cost, _, _, _ = R.compute_cost(p, cfg, feat, fm, lab, lm, None, 1)
    cost.backward()
    flat = torch.cat([p[n].grad.reshape(-1) for n in names]).float()
    assert abs(ret["cost"] - float(cost)) < 1e-5 * abs(float(cost))
    assert abs(ret["den"] - float(fm[1:].sum())) < 1e-6
    assert torch.allclose(ret["flat"], flat, rtol=1e-4, atol=1e-7)


def test_shard_batch_covers_batch():
    from parrot_amd.dist import shard_batch
    for B in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _worker_local_only(rank, world, port, ret):
    """bench.py's shape: rank 0 does extra single-rank work (the roofline leg) while the others wait at a barrier."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from parrot_amd import dist as pdist
    pdist.init_process_group(backend="gloo")
    x = torch.full((4,), float(rank + 1))
    if rank == 0:
        with pdist.local_only():
            assert not pdist.is_distributed()
            y = x.clone()
            pdist.allreduce_flat_(y)                      # must NOT talk to rank 1 (which is at the barrier below)
            pdist.broadcast_parameters_(y)
            scale, den = pdist.global_cost_scale(torch.tensor(3.0))
            assert torch.equal(y, x) and float(den) == 3.0
        assert pdist.is_distributed()
    pdist.barrier()
    pdist.allreduce_flat_(x)                              # the collectives are still in step afterwards
    if rank == 0:
        ret["sum"] = x.tolist()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_local_only_issues_no_collectives():
    world, port = 2, 29519
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_local_only, args=(world, port, ret), nprocs=world, join=True)
    assert ret["sum"] == [3.0, 3.0, 3.0, 3.0]


def _worker_buckets(rank, world, port, ret):
    """dist.GradientExchange: the early bucket issued asynchronously in the middle of 'the backward pass', the rest at the
    end, against one flat all-reduce of the same buffer -- and the bf16-on-the-wire option."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from parrot_amd import dist as pdist
    torch.set_num_threads(1)
    pdist.init_process_group(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)
    n, lo, hi = 10007, 4000, 6504
    grad = torch.randn(n, generator=g) * torch.logspace(-6, 3, n)  # magnitudes over nine decades
    flat = grad.clone()
    pdist.allreduce_flat_(flat)                       # reference: one bucket
    out = {}
    for name, early, fire in (("two_buckets", (lo, hi), True), ("hook_never_fired", (lo, hi), False),
                              ("no_early_range", None, True), ("bad_range_ignored", (hi, lo), True)):
        buf = grad.clone()
        ex = pdist.GradientExchange(buf, early)
        if fire:
            ex.start_early()
            ex.start_early()                          # a second call in the same step must not reduce twice
        buf[:lo].mul_(1.0)                            # ("the scan" keeps writing outside the early slice meanwhile)
        ex.finish()
        out[name] = bool(torch.equal(buf, flat))
        buf2 = grad.clone()                           # the exchange object is reusable: second step, same result
        ex.flat = buf2
        if fire:
            ex.start_early()
        ex.finish()
        out[name + "_again"] = bool(torch.equal(buf2, flat))
    # readiness order (round 6): the early bucket, then several ranges reported one by one in no particular address order
    # ("each deferred weight-gradient matrix as its product completes"), the gaps at the end -- still bit for bit
    buf = grad.clone()
    ex = pdist.GradientExchange(buf, (lo, hi))
    ex.start_early()
    for a_, b_ in ((8000, 9000), (0, 12), (6504, 7000), (9000, 10007)):
        ex.mark_ready(a_, b_)
    try:
        ex.mark_ready(8500, 8600)                     # handed over already: must raise, not double-sum
        out["overlap_raises"] = False
    except RuntimeError:
        out["overlap_raises"] = True
    ex.finish()
    out["ready_ranges"] = bool(torch.equal(buf, flat))
    # bf16 on the wire: each rank's gradient rounded once, summed, widened; compare with that arithmetic done by hand
    others = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(others, grad)
    want = sum(o.to(torch.bfloat16).float() for o in others).to(torch.bfloat16).float()
    buf = grad.clone()
    ex = pdist.GradientExchange(buf, (lo, hi), wire_dtype=torch.bfloat16)
    try:
        ex.start_early()
        ex.finish()
        out["bf16_wire"] = bool(torch.allclose(buf, want, rtol=1e-2, atol=0)) and buf.dtype == torch.float32
    except RuntimeError as e:  # a gloo build without bf16 reductions: say so instead of failing the f32 checks
        out["bf16_wire"] = "unsupported by this gloo: " + str(e)[:80]
    pdist.barrier()
    if rank == 0:
        ret.update(out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_gradient_exchange_equals_flat_allreduce_bit_for_bit():
    world, port = 2, 29523
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_buckets, args=(world, port, ret), nprocs=world, join=True)
    res = dict(ret)
    bf = res.pop("bf16_wire")
    assert res and all(res.values()), res
    assert bf is True or (isinstance(bf, str) and bf.startswith("unsupported")), bf


def test_gradient_exchange_is_a_no_op_outside_a_process_group():
    from parrot_amd import dist as pdist
    buf = torch.arange(10.0)
    ex = pdist.GradientExchange(buf, (2, 5))
    ex.start_early()
    ex.finish()
    assert torch.equal(buf, torch.arange(10.0))
