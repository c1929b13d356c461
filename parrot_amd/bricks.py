"""Blocks-style bricks backed by the HIP library.

Mirrors the brick surface model.py uses from the (un-vendored) Blocks package -- reference
model.py:1-9: ``Linear``, ``Fork``, ``LookupTable``, ``GatedRecurrent``, ``Bidirectional``,
``Initializable`` and the initialisation schemes of train.py:30-31 -- with the same names,
argument meaning and ``apply`` conventions (row-vector ``x . W + b``, ``W [in, out]``).

Every ``apply`` runs HIP kernels from libparrot_hip.so through parrot_amd.ops and is
differentiable with torch.autograd (custom Functions), so a brick can be used stand-alone.
Tensors must live on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import copy
import math
from collections import OrderedDict

import torch

from . import ops


# ----------------------------------------------------------------------------- initialisation
class NdarrayInitialization:
    def generate(self, gen: torch.Generator, shape):
        raise NotImplementedError

    def initialize(self, tensor: torch.Tensor, gen: torch.Generator):
        tensor.copy_(self.generate(gen, tuple(tensor.shape)).to(tensor.device, tensor.dtype))


class Constant(NdarrayInitialization):
    """blocks.initialization.Constant (train.py:31)."""

    def __init__(self, constant):
        self.constant = constant

    def generate(self, gen, shape):
        return torch.full(shape, float(self.constant), dtype=torch.float32)


class IsotropicGaussian(NdarrayInitialization):
    """blocks.initialization.IsotropicGaussian(std, mean) (train.py:30)."""

    def __init__(self, std=1.0, mean=0.0):
        self.std, self.mean = std, mean

    def generate(self, gen, shape):
        return (torch.randn(shape, generator=gen, dtype=torch.float64) * self.std + self.mean).float()


class Uniform(NdarrayInitialization):
    """Uniform with a given standard deviation (sampleRNN/lib/ops.py:19-30)."""

    def __init__(self, std=1.0):
        self.std = std

    def generate(self, gen, shape):
        a = self.std * math.sqrt(3.0)
        return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * a).float()


class Orthogonal(NdarrayInitialization):
    """Orthogonal init used for the candidate weights (sampleRNN/lib/ops.py:77-91)."""

    def generate(self, gen, shape):
        flat = (shape[0], int(math.prod(shape[1:])))
        a = torch.randn(flat, generator=gen, dtype=torch.float64)
        u, _, vh = torch.linalg.svd(a, full_matrices=False)
        q = u if u.shape == flat else vh
        return q.reshape(shape).float()


# ----------------------------------------------------------------------------- brick base
class Brick:
    """Minimal stand-in for blocks.bricks.Brick / Initializable: named parameter owner with
    children, a hierarchical path ("/parent/child.param") and initialize()."""

    def __init__(self, name=None, weights_init=None, biases_init=None, device=None, seed=None):
        self.name = name or type(self).__name__.lower()
        self.children = []
        self.parameters = OrderedDict()  # short name -> tensor
        self.roles = {}  # short name -> "weight" | "bias" | "initial_state"
        self.weights_init = weights_init
        self.biases_init = biases_init
        self.device = torch.device(device) if device is not None else None
        self.seed = seed
        self.initialized = False

    # -- parameters
    def _dev(self):
        if self.device is None:
            # Parameters may be *held* on the CPU (bookkeeping, checkpoint conversion); every
            # apply() still requires GPU tensors and raises otherwise (ops._chk).
            self.device = (torch.device("cuda", torch.cuda.current_device())
                           if torch.cuda.is_available() else torch.device("cpu"))
        return self.device

    def add_parameter(self, short, shape, role="weight", requires_grad=True):
        t = torch.zeros(shape, device=self._dev(), dtype=torch.float32, requires_grad=requires_grad)
        self.roles[short] = role
        self.parameters[short] = t
        return t

    def push_initialization_config(self):
        for c in self.children:
            if c.weights_init is None:
                c.weights_init = self.weights_init
            if c.biases_init is None:
                c.biases_init = self.biases_init
            if c.device is None:
                c.device = self.device
            c.push_initialization_config()

    def initialize(self, gen=None):
        """Initializable.initialize(): weights_init for weights, biases_init for biases; initial
        states start at zero (Blocks recurrent bricks)."""
        if gen is None:
            gen = torch.Generator().manual_seed(1 if self.seed is None else int(self.seed))
        self.push_initialization_config()
        self._initialize(gen)
        for c in self.children:
            c.initialize(gen)
        self.initialized = True
        return self

    def _initialize(self, gen):
        with torch.no_grad():
            for short, t in self.parameters.items():
                role = self.roles.get(short, "weight")
                if role == "weight" and self.weights_init is not None:
                    self.weights_init.initialize(t, gen)
                elif role == "bias" and self.biases_init is not None:
                    self.biases_init.initialize(t, gen)
                elif role == "initial_state":
                    t.zero_()

    def get_parameter_dict(self, prefix=""):
        """name -> tensor with Blocks-style hierarchical names (sample.py:83)."""
        path = f"{prefix}/{self.name}"
        out = OrderedDict()
        for short, t in self.parameters.items():
            out[f"{path}.{short}"] = t
        for c in self.children:
            out.update(c.get_parameter_dict(path))
        return out

    def all_parameters(self):
        return list(self.get_parameter_dict().values())


Initializable = Brick


# ----------------------------------------------------------------------------- feed-forward bricks
class Linear(Brick):
    """blocks.bricks.Linear: apply(x) = x . W + b."""

    def __init__(self, input_dim, output_dim, use_bias=True, **kw):
        super().__init__(**kw)
        self.input_dim, self.output_dim, self.use_bias = input_dim, output_dim, use_bias
        self.add_parameter("W", (input_dim, output_dim), "weight")
        if use_bias:
            self.add_parameter("b", (output_dim,), "bias")

    @property
    def W(self):
        return self.parameters["W"]

    @property
    def b(self):
        return self.parameters.get("b")

    def apply(self, x):
        return ops.linear(x, self.W, self.b)


class Fork(Brick):
    """blocks.bricks.parallel.Fork: one independent Linear per output name (children named
    ``fork_<output_name>``); apply returns them in output_names order."""

    def __init__(self, output_names, input_dim, output_dims, **kw):
        super().__init__(**kw)
        self.output_names, self.input_dim, self.output_dims = list(output_names), input_dim, list(output_dims)
        self.children = [Linear(input_dim, od, name=f"fork_{on}") for on, od in zip(output_names, output_dims)]

    def apply(self, x, as_dict=False):
        outs = [c.apply(x) for c in self.children]
        if as_dict:
            return OrderedDict(zip(self.output_names, outs))
        return outs


class LookupTable(Brick):
    """blocks.bricks.lookup.LookupTable: apply(idx) = W[idx]."""

    def __init__(self, length, dim, **kw):
        super().__init__(**kw)
        self.length, self.dim = length, dim
        self.add_parameter("W", (length, dim), "weight")

    @property
    def W(self):
        return self.parameters["W"]

    def apply(self, indices):
        return self.W[indices.long()]


# ----------------------------------------------------------------------------- recurrent bricks
class GatedRecurrent(Brick):
    """blocks.bricks.recurrent.GatedRecurrent (GRU without biases; biases live in the Forks).

    apply(inputs, gate_inputs, states=None, mask=None, iterate=True):
      iterate=False: one step on [B, dim] tensors (model.py:659-662);
      iterate=True : scan over axis 0 of [T, B, .] tensors, returns all states [T, B, dim].
    """

    sequences = ["mask", "inputs", "gate_inputs"]
    states = ["states"]

    def __init__(self, dim, **kw):
        super().__init__(**kw)
        self.dim = dim
        self.add_parameter("state_to_state", (dim, dim), "weight")
        self.add_parameter("state_to_gates", (dim, 2 * dim), "weight")
        self.add_parameter("initial_state", (dim,), "initial_state")

    def get_dim(self, name):
        if name in ("inputs", "states"):
            return self.dim
        if name == "gate_inputs":
            return 2 * self.dim
        if name == "mask":
            return 0
        raise ValueError(name)

    def initial_states(self, batch_size):
        return self.parameters["initial_state"].unsqueeze(0).expand(batch_size, -1)

    def apply(self, inputs, gate_inputs, states=None, mask=None, iterate=True, reverse=False):
        Wc, Wg = self.parameters["state_to_state"], self.parameters["state_to_gates"]
        if not iterate:
            if states is None:
                states = self.initial_states(inputs.shape[0])
            return ops.gru_step(inputs, gate_inputs, states, Wc, Wg, mask)
        if states is None:
            states = self.initial_states(inputs.shape[1])
        return ops.gru_seq(inputs, gate_inputs, states, Wc, Wg, mask, reverse)


class RecurrentWithFork(Brick):
    """model.py:172-198: a recurrent transition whose sequence inputs come from a Fork."""

    def __init__(self, recurrent: GatedRecurrent, input_dim, **kw):
        super().__init__(**kw)
        self.recurrent = recurrent
        self.input_dim = input_dim
        names = [n for n in recurrent.sequences if n != "mask"]
        self.fork = Fork(names, input_dim, [recurrent.get_dim(n) for n in names], name="fork")
        self.children = [recurrent, self.fork]

    def apply(self, input_, mask=None, reverse=False):
        inp, gat = self.fork.apply(input_)
        return self.recurrent.apply(inp, gat, mask=mask, reverse=reverse)


class Bidirectional(Brick):
    """blocks.bricks.recurrent.Bidirectional: two independent copies of the prototype named
    ``forward`` / ``backward``; apply = concat([fwd(x), bwd(x[::-1])[::-1]], axis=-1).
    Recurrences iterate axis 0."""

    def __init__(self, prototype, **kw):
        super().__init__(**kw)
        self.prototype = prototype
        fwd, bwd = copy.deepcopy(prototype), copy.deepcopy(prototype)
        fwd.name, bwd.name = "forward", "backward"
        self.children = [fwd, bwd]

    def apply(self, x, mask=None):
        f = self.children[0].apply(x, mask)
        b = self.children[1].apply(x, mask, reverse=True)
        return torch.cat([f, b], dim=-1)
