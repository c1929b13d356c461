"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The reference is single-device (no NCCL/MPI anywhere); data parallelism attaches
at the gradient step (train.py:100-108).  The batch is sharded along B only (utterances are
independent through the whole step); two things couple the shards and both are handled here:

  * the masked-mean cost denominator (model.py:784) is over the GLOBAL batch, so every rank scales its
    backward pass by (den_local + eps) / (den_global + eps) before the gradient all-reduce;
  * StepClipping uses the global-norm of the SUMMED gradient (train.py:100-101), so the norm is
    taken after the all-reduce.

Gradients live in one flat float32 buffer (parrot_amd.params.ParamStore): 55.7 MB for BASELINE cfg2, 100.8 MB for the
3-layer model, 277.8 MB for configs[3].  `GradientExchange` sums it in two buckets: the readout / output share, complete
before the backward scan starts, travels while the scan runs; the rest follows when the backward pass ends.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

COST_EPS = 1e-5  # model.py:784


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _force_one_rank():
    """PARROT_DIST_FORCE=1: a ONE-rank process group counts as distributed -- every collective of the training step is issued
    (and is the identity).  The only way to drive the RCCL path of Trainer.step on a box with a single GPU
    (tests/test_gpu_dp.py::test_one_rank_rccl_trainer_step_equals_plain_step)."""
    return os.environ.get("PARROT_DIST_FORCE", "0") == "1"


def init_process_group(backend=None):
    rank, local_rank, world = env_world()
    if (world > 1 or _force_one_rank()) and not dist.is_initialized():
        if backend is None:
            # PARROT_DIST_BACKEND=gloo: plumbing test of the N > 1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("PARROT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


_LOCAL_ONLY = 0


class local_only:
    """Context manager: inside it this rank behaves like a single process (no collective is issued), e.g. for a
    measurement that only rank 0 performs while the other ranks wait at a barrier."""

    def __enter__(self):
        global _LOCAL_ONLY
        _LOCAL_ONLY += 1
        return self

    def __exit__(self, *exc):
        global _LOCAL_ONLY
        _LOCAL_ONLY -= 1
        return False


def is_distributed():
    if _LOCAL_ONLY:
        return False
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force_one_rank())


def shard_batch(batch_size, rank, world):
    """Contiguous shard [lo, hi) of the global batch for this rank (remainder to the low ranks)."""
    base, rem = divmod(batch_size, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_cost_scale(den_local: torch.Tensor, group=None):
    """Returns (scale, den_global): scale = (den_local + eps) / (den_global + eps) turns the gradient of
    the rank-local masked mean into this rank's share of the global masked mean's gradient."""
    den_global = den_local.detach().clone().reshape(1).to(torch.float32)
    if is_distributed():
        dist.all_reduce(den_global, op=dist.ReduceOp.SUM, group=group)
    scale = (den_local + COST_EPS) / (den_global + COST_EPS)
    return scale.reshape(()), den_global.reshape(())


def allreduce_flat_(flat: torch.Tensor, group=None, async_op=False):
    """Sum-all-reduce of the flat gradient bucket, in place."""
    if not is_distributed():
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class GradientExchange:
    """The gradient sum of one training step over one flat buffer, in up to two buckets (round 5).

    `early = (lo, hi)`: the slice of the flat gradient that is complete BEFORE the backward scan starts -- the readout stack
    and the output layer (model.py:739-755; `Parrot.early_gradient_range()`).  `start_early()` is called at that moment and
    issues its all-reduce asynchronously (RCCL runs it on its own stream behind what the compute stream holds so far), so it
    travels over xGMI while the backward scan and the deferred weight-gradient products run; `finish()` reduces what is left
    (the recurrent / attention / encoder gradients, complete only when the backward pass ends) and waits for the early
    bucket.  Every element is summed exactly once by exactly one collective, so at world size 2 the result equals the
    one-bucket all-reduce bit for bit (tests/test_dist_cpu.py).

    `wire_dtype=torch.bfloat16` (opt-in; SURVEY.md K16 budgets the configs[3] exchange at 138.9 MB in bf16 against 277.8 MB
    in f32): every rank rounds its f32 gradient to bf16 once, the collective sums bf16 values, the result is widened back
    into the f32 buffer -- the master gradients, the clip norm and Adam stay f32.  Changes the arithmetic (one rounding per
    rank and per hop of the collective), so it is a flag, not the default."""

    def __init__(self, flat: torch.Tensor, early=None, wire_dtype=None, group=None):
        self.flat, self.group = flat, group
        self.wire = wire_dtype if wire_dtype not in (None, torch.float32) else None
        n = flat.numel()
        if early is not None:
            lo, hi = int(early[0]), int(early[1])
            if not (0 <= lo < hi <= n):
                early = None
            else:
                early = (lo, hi)
        self.early = early
        self._pending = []   # (work, wire tensor or None, destination view)
        self._ready = []     # [lo, hi) ranges handed over this step
        self._early_done = False

    def _issue(self, view, async_op):
        if view.numel() == 0:
            return
        if self.wire is None:
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self._pending.append((w, None, view))
            return
        t = view.to(self.wire)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self._pending.append((w, t, view))
        else:
            view.copy_(t)

    def mark_ready(self, lo, hi):
        """flat[lo:hi] is final for this step: its all-reduce is issued now, asynchronously (round 6: the deferred weight-
        gradient products report their outputs one by one, so each bucket travels beside the products that follow it and
        only the last one's bytes are exposed).  Ranges must not overlap within a step; a no-op outside a process group.

        Contract: ONE backward pass per `finish()`.  A slice that has been handed over is being summed in place -- writing
        to it again before `finish()` (a second backward under gradient accumulation, a parameter that receives a later
        contribution) is lost or double-counted; an overlapping range raises."""
        if not is_distributed():
            return
        lo, hi = int(lo), int(hi)
        if not (0 <= lo < hi <= self.flat.numel()):
            raise ValueError(f"mark_ready: [{lo}, {hi}) outside the buffer")
        for a, b in self._ready:
            if lo < b and a < hi:
                raise RuntimeError(f"mark_ready: [{lo}, {hi}) overlaps [{a}, {b}), already handed over this step")
        self._ready.append((lo, hi))
        self._issue(self.flat[lo:hi], async_op=True)

    def start_early(self):
        """Call when flat[early] is final for this step (at most once per step; a no-op outside a process group)."""
        if not is_distributed() or self.early is None or self._early_done:
            return
        self._early_done = True
        self.mark_ready(*self.early)

    def finish(self):
        """Reduces whatever was not handed over by `mark_ready` / `start_early` (the gaps between the ready ranges, in
        address order) and waits for everything; the buffer then holds the global sum."""
        if not is_distributed():
            self._early_done = False
            self._ready = []
            return
        pos = 0
        for a, b in sorted(self._ready):
            self._issue(self.flat[pos:a], async_op=False)
            pos = b
        self._issue(self.flat[pos:], async_op=False)
        for w, t, view in self._pending:
            w.wait()
            if t is not None:
                view.copy_(t)
        self._pending = []
        self._ready = []
        self._early_done = False


def allreduce_cost(num_local: torch.Tensor, den_global: torch.Tensor, group=None):
    """Global masked-mean cost from the rank-local numerators."""
    num = num_local.detach().clone().reshape(1).to(torch.float32)
    if is_distributed():
        dist.all_reduce(num, op=dist.ReduceOp.SUM, group=group)
    return (num / (den_global + COST_EPS)).reshape(())


def broadcast_parameters_(flat: torch.Tensor, src=0, group=None):
    if is_distributed():
        dist.broadcast(flat, src=src, group=group)


def barrier():
    if is_distributed():
        dist.barrier()


def any_rank(flag: bool, device=None) -> bool:
    """Collective OR: every rank gets True when any rank passed True (stop / time-limit decisions must be the same
    everywhere, or the ranks that continue hang in the next gradient all-reduce)."""
    if not is_distributed():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device or _coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def sum_over_ranks(values, device=None):
    """Sum-all-reduce of a few python floats (validation numerator / denominator); returns a list of floats."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device or _coll_device())
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def _coll_device():
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')
