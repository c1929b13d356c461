"""Flat parameter / gradient storage with named views.

One float32 buffer holds every parameter of a model and a second one every gradient, so that
  * the data-parallel gradient all-reduce is a single RCCL call on one bucket (train.py:103-108 is
    where it attaches; the reference itself is single-device),
  * the fused clip + Adam kernel walks one array,
  * the packed per-layer weight matrices the scan kernels want and the reference-named parameters
    (Blocks brick paths, sample.py:83) are both plain views of the same memory.
"""
from __future__ import annotations

from collections import OrderedDict

import torch


class ParamStore:
    def __init__(self):
        self._entries = OrderedDict()   # storage name -> (shape, role)
        self._aliases = OrderedDict()   # reference name -> (storage name, row_slice, col_slice, role)
        self.flat = None
        self.flat_grad = None
        self.storage = OrderedDict()    # storage name -> view of flat
        self.storage_grad = OrderedDict()
        self.offsets = {}
        self._transposed = set()

    # -- declaration
    def add(self, name, shape, role="weight", alias=True):
        assert name not in self._entries, name
        self._entries[name] = (tuple(shape), role)
        if alias:
            self._aliases[name] = (name, None, None, role)

    def alias(self, refname, storage, rows=None, cols=None, role="weight", transpose=False):
        assert refname not in self._aliases, refname
        self._aliases[refname] = (storage, rows, cols, role)
        if transpose:
            self._transposed.add(refname)

    # -- allocation
    def allocate(self, device):
        off = 0
        for name, (shape, _) in self._entries.items():
            n = 1
            for s in shape:
                n *= s
            self.offsets[name] = (off, n)
            off += (n + 3) // 4 * 4  # keep every entry 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=device, dtype=torch.float32)
        for name, (shape, _) in self._entries.items():
            o, n = self.offsets[name]
            self.storage[name] = self.flat[o:o + n].view(shape)
            self.storage_grad[name] = self.flat_grad[o:o + n].view(shape)
        return self

    def _view(self, table, refname):
        storage, rows, cols, _ = self._aliases[refname]
        t = table[storage]
        if rows is not None:
            t = t[rows[0]:rows[1]]
        if cols is not None:
            t = t[..., cols[0]:cols[1]]
        if refname in self._transposed:
            t = t.t()
        return t

    def param(self, refname):
        return self._view(self.storage, refname)

    def grad(self, refname):
        return self._view(self.storage_grad, refname)

    def role(self, refname):
        return self._aliases[refname][3]

    def names(self):
        return list(self._aliases.keys())

    def named_parameters(self):
        return OrderedDict((n, self.param(n)) for n in self._aliases)

    def named_gradients(self):
        return OrderedDict((n, self.grad(n)) for n in self._aliases)

    def zero_grad(self):
        self.flat_grad.zero_()
