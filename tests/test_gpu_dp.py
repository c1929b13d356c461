"""The PRODUCT through the data-parallel path: two ranks on one GPU (gloo carries the
collectives, PARROT_DIST_BACKEND=gloo), each running Trainer.step on its half of the batch, must leave the same
parameters as one process stepping on the whole batch: global masked-mean denominator, flat gradient all-reduce,
global-norm clip after the reduction, Adam on identical replicas.  No 1 -> 8 GPU scaling curve is measured here or
anywhere in this repository's own runs (gpurun exposes one GPU; the driver runs SCALE_rNN)."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests.util import make_batch

pytestmark = pytest.mark.gpu

KW = dict(num_layers=2, rnn_h_dim=64, readouts_dim=48, encoder_dim=16, input_dim=24, encoder_type='bidirectional',
          weak_feedback=True, encoder_literal=False)
T, B, U = 9, 6, 7


def _step(dev, lo, hi, steps=2):
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    from parrot_amd.trainer import Trainer
    cfg = R.default_config(**KW)
    p = R.init_params(cfg, seed=7, scale_by_fan_in=True)
    m = Parrot(device=dev, use_graph=True, **KW).allocate()
    m.set_parameter_values(p)
    tr = Trainer(m, learning_rate=1e-2, grad_clip=0.05)  # small threshold: the clip is active
    feat, fm, lab, lm, _ = make_batch(cfg, T, B, U, seed=3, ragged=True)
    costs = []
    for s in range(steps):  # second step: start_flag = 0, carried state stays rank-local
        a, b = (0, 5) if s == 0 else (4, T)
        c = tr.step(feat[a:b + 1, lo:hi].float().to(dev), fm[a:b + 1, lo:hi].float().to(dev), lab[lo:hi].to(dev),
                    lm[lo:hi].float().to(dev), None, 1 if s == 0 else 0)
        costs.append(float(c))
    out = (m.flat_parameters.detach().cpu().clone(), costs, float(tr.gnorm_sq))
    m.close()
    return out


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PARROT_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from parrot_amd import dist as pdist
    pdist.init_process_group()
    assert pdist.is_distributed()
    lo, hi = pdist.shard_batch(B, rank, world)
    flat, costs, gn = _step(torch.device("cuda:0"), lo, hi)
    ret[rank] = (flat, costs, gn)
    pdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_trainer_step_equals_single_process(dev):
    world, port = 2, 29533
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    flat1, costs1, gn1 = _step(dev, 0, B)
    for r in range(world):
        flat, costs, gn = ret[r]
        assert abs(gn - gn1) <= 1e-4 * abs(gn1), "global gradient norm (after the all-reduce)"
        for a, b in zip(costs, costs1):
            assert abs(a - b) <= 1e-5 * abs(b), "global cost"
        err = float((flat - flat1).abs().max() / flat1.abs().max())
        assert err <= 1e-5, f"rank {r}: parameters after two clip+Adam steps differ by {err:.2e}"
    assert torch.equal(ret[0][0], ret[1][0]), "replicas stay bit-identical"


def _worker_rccl(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      PARROT_DIST_BACKEND="nccl", PARROT_DIST_FORCE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from parrot_amd import dist as pdist
    pdist.init_process_group()
    assert pdist.is_distributed() and dist.get_backend() == "nccl"
    flat, costs, gn = _step(torch.device("cuda:0"), 0, B)
    ret[0] = (flat, costs, gn)
    pdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_rccl_trainer_step_equals_plain_step(dev):
    """Trainer.step through the distributed path on RCCL itself (backend "nccl", a one-rank group is
    what one GPU allows): the mask-count all-reduce, the early bucket handed over by the backward pass while the scan's
    hipGraph replays, every deferred weight-gradient matrix handed over by mark_ready, the gaps in finish(), the cost
    reduction -- asynchronous RCCL collectives on their own stream beside the compute stream.  With one rank every sum is
    the identity, so the parameters after two clip + Adam steps must equal the non-distributed step's bit for bit."""
    ret = mp.Manager().dict()
    mp.spawn(_worker_rccl, args=(29541, ret), nprocs=1, join=True)
    flat1, costs1, gn1 = _step(dev, 0, B)
    flat, costs, gn = ret[0]
    assert costs == costs1 and gn == gn1
    assert torch.equal(flat, flat1), "one-rank RCCL step differs from the plain step"


@pytest.mark.parametrize("kw", [dict(), dict(which_cost='GMM', k_gmm=3), dict(layer_norm=True), dict(use_speaker=True, num_speakers=4),
                                dict(cell_type='lstm', num_layers=3), dict(num_layers=1, weak_feedback=False)])
def test_early_gradient_bucket_is_final_when_the_hook_fires(dev, kw):
    """dist.GradientExchange starts the all-reduce of `Parrot.early_gradient_range()` the moment the backward pass calls
    `on_early_gradients` -- before the backward scan.  What it promises: nothing in the rest of the backward pass writes that
    slice of the flat gradient again.  Checked here for every head / variant: the slice captured inside the hook equals the
    slice after the whole backward, bit for bit; it is not empty, holds non-zero gradients, starts at the readout matrix and
    ends with the output layer; everything outside it is still changing when the hook fires (the split is worth something)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    full = dict(KW, **kw)
    cfg = R.default_config(**full)
    p = R.init_params(cfg, seed=11, scale_by_fan_in=True)
    m = Parrot(device=dev, use_graph=True, **full).allocate()
    m.set_parameter_values(p)
    feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=5, ragged=True, speaker=cfg['use_speaker'])
    rng = m.early_gradient_range()
    assert rng is not None
    lo, hi = rng
    names = list(m.store._entries)
    assert lo == m.store.offsets['dec.Wr'][0] and 0 <= lo < hi <= m.flat_gradients.numel()
    seen = []
    m.on_early_gradients = lambda: seen.append((m.flat_gradients[lo:hi].clone(), m.flat_gradients.clone()))
    m.zero_grad()
    cost, _, _, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev),
                                   None if spk is None else spk.to(dev), 1, B)
    cost.backward()
    m.on_early_gradients = None
    assert len(seen) == 1, "the hook fires exactly once per backward pass"
    early_then, all_then = seen[0]
    final = m.flat_gradients
    assert torch.equal(early_then, final[lo:hi]), "the early bucket was written again after the hook"
    assert float(early_then.abs().max()) > 0.0
    outside_then = torch.cat([all_then[:lo], all_then[hi:]])
    outside_final = torch.cat([final[:lo], final[hi:]])
    assert not torch.equal(outside_then, outside_final), "nothing was left to do after the hook: the split buys nothing"
    # the output layer's parameters are inside the bucket
    for wn, bn, _ in m._out_names:
        for n in (wn, bn):
            o, k = m.store.offsets[n]
            assert lo <= o and o + k <= hi, n
    m.close()
