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
