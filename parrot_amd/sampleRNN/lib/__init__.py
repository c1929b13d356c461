"""Parameter registry of the SampleRNN code base -- mirrors sampleRNN/lib/__init__.py:28-47 (`param`),
:84-94 (`floatX`) and :96-109 (`save_params` / `load_params`).

`param(name, value)` creates a parameter the first time a name is seen and returns the SAME tensor on
every later call, which is how the reference shares weights between the training graph and the three
generation functions.  Checkpoints are pickled {name: ndarray} dicts with the reference's dotted names
(``BigFrameLevel.GRU1.Step.Input.W0``, ``FrameLevel.h0``, ``SampleLevel.Embedding`` ...).
The run bookkeeping / plotting helpers of the reference (:127-387) belong to its commented-out
training loop and are out of scope.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict

import numpy
import torch

_params = OrderedDict()
_device = None


def set_device(device):
    global _device
    _device = torch.device(device)


def device():
    global _device
    if _device is None:
        _device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device('cpu')
    return _device


def param(name, value=None, trainable=True):
    """lib.param (lib/__init__.py:28-47)."""
    if name not in _params:
        if value is None:
            raise KeyError(f'parameter {name} does not exist yet and no initial value was given')
        t = torch.as_tensor(numpy.asarray(value, dtype='float32') if not isinstance(value, torch.Tensor) else value)
        t = t.to(device(), torch.float32).clone().requires_grad_(trainable)
        t.param = trainable
        t.name_ = name
        _params[name] = t
    return _params[name]


def delete_all_params():
    _params.clear()


def get_params(predicate=None):
    return [p for n, p in _params.items() if predicate is None or predicate(n, p)]


def named_params():
    return OrderedDict(_params)


def floatX(x):
    """lib.floatX (lib/__init__.py:84-94)."""
    return numpy.float32(x)


def save_params(path):
    """lib/__init__.py:96-102: pickle of {name: ndarray}."""
    blob = {n: p.detach().cpu().numpy() for n, p in _params.items()}
    with open(path, 'wb') as f:
        pickle.dump(blob, f, protocol=2)


def load_params(path):
    """lib/__init__.py:104-109."""
    with open(path, 'rb') as f:
        blob = pickle.load(f, encoding='latin1')
    set_params(blob)


def set_params(values):
    with torch.no_grad():
        for n, v in values.items():
            v = torch.as_tensor(numpy.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, dtype='float32'))
            if n in _params:
                _params[n].copy_(v.to(_params[n].device))
            else:
                param(n, v)


def flatten_params():
    """Moves every registered parameter into one flat float32 buffer (and its gradient into a second one),
    keeping the registry objects' identity semantics: `param(name)` now returns a leaf view of the flat
    buffer whose `.grad` is a view of the flat gradient buffer.  Returns (flat, flat_grad).  This is what
    the fused clip+Adam kernel and the data-parallel all-reduce operate on."""
    names = list(_params.keys())
    sizes = [(_params[n].numel() + 3) // 4 * 4 for n in names]
    total = sum(sizes)
    dev = device()
    flat = torch.zeros(total, device=dev, dtype=torch.float32)
    flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
    off = 0
    for n, sz in zip(names, sizes):
        old = _params[n]
        k = old.numel()
        with torch.no_grad():
            flat[off:off + k].copy_(old.detach().reshape(-1))
        view = flat[off:off + k].view(old.shape).detach()
        trainable = bool(getattr(old, 'param', True))
        view.requires_grad_(trainable)
        if trainable:
            view.grad = flat_grad[off:off + k].view(old.shape)
        view.param = trainable
        view.name_ = n
        _params[n] = view
        off += sz
    return flat, flat_grad
