"""Training step next to the hot path: StepClipping(10*grad_clip) o Adam(lr) on the flat buffers
(reference train.py:100-108, Blocks defaults), with the data-parallel gradient exchange in between.

Replaces Blocks' GradientDescent/MainLoop machinery (out of scope) with a
plain loop; the optional patience / halve-LR schedule of extensions.py:83-152 is kept because it
changes the arithmetic of the optimiser state."""
from __future__ import annotations

import torch

from . import dist as pdist
from . import ops


class Trainer:
    def __init__(self, parrot, learning_rate=1e-4, grad_clip=0.9, beta1=0.9, beta2=0.999, eps=1e-8,
                 bucketed=None, allreduce_dtype=None):
        """bucketed (default: PARROT_DP_BUCKETS != 0): the decoder's gradient sum in two buckets, the readout / output share
        overlapped with the backward scan (dist.GradientExchange).  allreduce_dtype='bf16' (default: PARROT_ALLREDUCE_BF16):
        bf16 on the wire, f32 master gradients (opt-in; changes the arithmetic)."""
        import os
        self.parrot = parrot.allocate()
        if bucketed is None:
            bucketed = os.environ.get('PARROT_DP_BUCKETS', '1') != '0'
        if allreduce_dtype is None and os.environ.get('PARROT_ALLREDUCE_BF16', '0') != '0':
            allreduce_dtype = 'bf16'
        wire = torch.bfloat16 if allreduce_dtype in ('bf16', 'bfloat16', torch.bfloat16) else None
        self._wire = wire
        self.lr = float(learning_rate)
        self.clip = 10.0 * float(grad_clip)  # train.py:100-101: "for adam is 10x"
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.groups = [(parrot.flat_parameters, parrot.flat_gradients)]
        if getattr(parrot, 'raw_output', False):
            from .sampleRNN import lib as srn_lib
            flat, flat_grad = srn_lib.flatten_params()  # SampleRNN head: same clip + Adam, same global norm
            parrot.sampleRnn.parameters = srn_lib.get_params(lambda n, p_: getattr(p_, 'param', False))
            self.groups.append((flat, flat_grad))
        early = parrot.early_gradient_range() if (bucketed and hasattr(parrot, 'early_gradient_range')) else None
        self.exchanges = [pdist.GradientExchange(g_, early if i == 0 else None, wire)
                          for i, (_, g_) in enumerate(self.groups)]
        self.ms = [torch.zeros_like(p_) for p_, _ in self.groups]
        self.vs = [torch.zeros_like(p_) for p_, _ in self.groups]
        self.m, self.v = self.ms[0], self.vs[0]
        self._part = torch.zeros(1, device=parrot.flat_parameters.device, dtype=torch.float32)
        self.gnorm_sq = torch.zeros(1, device=parrot.flat_parameters.device, dtype=torch.float32)
        self.step_count = 0
        self.last_grad_norm = None
        for p_, _ in self.groups:  # identical replicas: every parameter group starts from rank 0's values
            pdist.broadcast_parameters_(p_)

    def step(self, features, features_mask, labels, labels_mask, speaker=None, start_flag=1,
             feedback_noise=None, raw_audio=None):
        """One training step on this rank's shard.  Returns the (global) cost as a 0-dim tensor."""
        p = self.parrot
        p.zero_grad()
        for _, g_ in self.groups[1:]:
            g_.zero_()
        B = features_mask.shape[1]
        cost, updates, _, _ = p.compute_cost(features, features_mask, labels, labels_mask, speaker,
                                             start_flag, B, raw_audio=raw_audio, feedback_noise=feedback_noise)
        den_local = features_mask[1:].to(cost.device, torch.float32).sum()
        if pdist.is_distributed():
            scale, den_global = pdist.global_cost_scale(den_local)
            p.on_early_gradients = self.exchanges[0].start_early  # fired by the backward pass (model.py side)
            p.on_gradient_ready = self.exchanges[0].mark_ready    # every deferred weight-gradient matrix as it completes
            try:
                cost.backward(gradient=scale.to(cost.dtype))
            finally:
                p.on_early_gradients = None
                p.on_gradient_ready = None
            for ex in self.exchanges:
                ex.finish()
            gcost = pdist.allreduce_cost(cost.detach() * (den_local + pdist.COST_EPS), den_global)
        else:
            cost.backward()
            gcost = cost.detach()
        p.apply_updates(updates)  # TBPTT carry (model.py:786-791)
        ops.sumsq(p.flat_gradients, out=self.gnorm_sq)
        for _, g_ in self.groups[1:]:  # StepClipping norm is over ALL parameters (train.py:100-101)
            ops.sumsq(g_, out=self._part)
            self.gnorm_sq.add_(self._part)
        self.step_count += 1
        for (p_, g_), m_, v_ in zip(self.groups, self.ms, self.vs):
            ops.adam_clip_step(p_, g_, m_, v_, self.gnorm_sq, self.step_count, lr=self.lr, clip=self.clip,
                               beta1=self.beta1, beta2=self.beta2, eps=self.eps)
        self.last_grad_norm = self.gnorm_sq
        return gcost

    # -- extensions.py:83-152 (LearningRateSchedule) arithmetic: halve LR, zero Adam buffers
    def cut_learning_rate(self, factor=0.5):
        self.lr *= factor
        for m_, v_ in zip(self.ms, self.vs):
            m_.zero_()
            v_.zero_()
        self.step_count = 0

    def state_dict(self):
        """Adam moments of EVERY parameter group (decoder + SampleRNN head), step count and learning rate."""
        return dict(ms=[m_.clone() for m_ in self.ms], vs=[v_.clone() for v_ in self.vs], step=self.step_count,
                    lr=self.lr)

    def load_state_dict(self, sd):
        ms = sd['ms'] if 'ms' in sd else [sd['m']]
        vs = sd['vs'] if 'vs' in sd else [sd['v']]
        assert len(ms) == len(self.ms), "checkpoint and trainer disagree on the number of parameter groups"
        for dst, src in zip(self.ms + self.vs, list(ms) + list(vs)):
            dst.copy_(src)
        self.step_count, self.lr = int(sd['step']), float(sd['lr'])


class LearningRateSchedule:
    """extensions.py:83-152 (wired at train.py:175-182 with patience=10, num_cuts=5, cut_size=.5): tracks the
    validation cost every `save_every` iterations; after `patience` checks without a new best (or at once on NaN)
    it reloads the best parameters, zeroes the optimiser buffers, multiplies the learning rate by `cut_size`, and
    asks for the end of training after `num_cuts` cuts.

    `update(value)` returns (cut, finish).  The caller supplies `reload_best()` (loads best_<exp>.tar into the model
    on every rank) -- the Blocks main-loop plumbing is not part of the arithmetic."""

    def __init__(self, trainer, reload_best, patience=5, num_cuts=3, cut_size=.5):
        self.trainer, self.reload_best = trainer, reload_best
        self.patience, self.num_cuts, self.cut_size = patience, num_cuts, cut_size
        self.counter = self.count_cuts = 0
        self.best_value = float('inf')

    def update(self, current_value):
        if current_value is None:
            return False, False
        current_value = float(current_value)
        if current_value < self.best_value:
            self.best_value, self.counter = current_value, 0
        else:
            self.counter += 1
        if current_value != current_value:  # NaN: skip the remaining patience (extensions.py:126-128)
            self.counter = self.patience + 1
        if self.counter < self.patience:
            return False, False
        self.counter = 0
        self.count_cuts += 1
        self.reload_best()
        self.trainer.cut_learning_rate(self.cut_size)
        return True, self.count_cuts >= self.num_cuts
