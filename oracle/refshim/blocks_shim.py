"""Minimal stand-in for the Blocks bricks that /root/reference/model.py instantiates (TEST INFRASTRUCTURE).

Blocks (mila-udem/blocks, un-vendored and un-pinned in the reference; API usage implies >= 0.2.0) cannot be installed
here.  This module restates the published behaviour of exactly the classes model.py imports (model.py:1-9):

  Brick / Initializable / Random   naming, children, allocate(); parameters are zero-filled shared variables that the
                                   fixture generator overwrites by hierarchical path (`/parrot/rnn1.state_to_state`),
                                   so Blocks' initialisation schemes are not restated
  Linear          apply(x) = x . W + b                       W [input_dim, output_dim], b [output_dim]
  Fork            one Linear child `fork_<output_name>` per output, apply -> list in output_names order
  LookupTable     apply(idx) = W[idx.flatten()].reshape(idx.shape + (dim,))
  GatedRecurrent  parameters state_to_state [d,d], state_to_gates [d,2d], initial_state [d];
                  gates = sigmoid(h . W_sg + gate_inputs); z = gates[:, :d]; r = gates[:, d:];
                  h' = tanh((h*r) . W_ss + inputs) * z + h * (1 - z); optional mask blend;
                  iterate=True scans over axis 0 of the sequences starting from initial_states(sequences.shape[1])
                  (`reverse=True` walks the sequences backwards, outputs in scan order)
  Bidirectional   two deep copies of the prototype named forward / backward; backward outputs are flipped and
                  concatenated on axis 2
  application / lazy decorators, as_dict / as_list call conventions, shared_floatx_zeros, dict_union, add_role.

The GatedRecurrent algebra is cross-checked against the reference's in-repo twin sampleRNN/lib/ops.py:364-393
(tests/test_oracle_cpu.py::test_blocks_shim_gru_equals_reference_grustep)."""
from __future__ import annotations

import collections
import copy
import functools
import types

import numpy

from . import theano_shim as th


# ------------------------------------------------------------------ application machinery
class Application:
    def __init__(self, fn, **meta):
        self.fn = fn
        self.meta = dict(meta)
        self.props = {}
        functools.update_wrapper(self, fn)

    def property(self, name):
        def deco(f):
            self.props[name] = f
            return f
        return deco

    def __get__(self, brick, owner=None):
        return self if brick is None else BoundApplication(self, brick)


class BoundApplication:
    def __init__(self, app, brick):
        self.application, self.brick = app, brick

    def _meta(self, name, default=None):
        app = self.application
        if name in app.props:
            return app.props[name](self.brick)
        return app.meta.get(name, default)

    @property
    def sequences(self):
        return self._meta('sequences', [])

    @property
    def states(self):
        return self._meta('states', [])

    @property
    def outputs(self):
        return self._meta('outputs', None)

    def __call__(self, *args, **kwargs):
        as_dict = kwargs.pop('as_dict', False)
        as_list = kwargs.pop('as_list', False)
        out = self.application.fn(self.brick, *args, **kwargs)
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        if as_dict:
            names = self.outputs
            return collections.OrderedDict(zip(names, outs))
        if as_list:
            return outs
        return outs[0] if len(outs) == 1 else outs

    def __deepcopy__(self, memo):
        return BoundApplication(self.application, copy.deepcopy(self.brick, memo))


def application(*args, **kwargs):
    if args and callable(args[0]) and not kwargs:
        return Application(args[0])
    return lambda f: Application(f, **kwargs)


def lazy(allocation=None, initialization=None):
    return lambda init: init


# ------------------------------------------------------------------ bricks
class Brick:
    def __init__(self, name=None, children=None, **kwargs):
        self.name = name if name is not None else type(self).__name__.lower()
        self.children = list(children) if children else []
        self.parameters = []
        self.allocated = False

    def _push_allocation_config(self):
        pass

    def _allocate(self):
        pass

    def push_allocation_config(self):
        self._push_allocation_config()
        for c in self.children:
            c.push_allocation_config()

    def allocate(self):
        self.push_allocation_config()
        self._allocate_tree()

    def _allocate_tree(self):
        for c in self.children:
            c._allocate_tree()
        if not self.allocated:
            self._allocate()
            self.allocated = True

    def initialize(self):
        if not self.allocated:
            self.allocate()

    def push_initialization_config(self):
        pass

    def named_parameters(self, prefix=''):
        """path -> shared variable, Blocks' hierarchical naming (`/parent/child.param`)."""
        path = f"{prefix}/{self.name}"
        out = collections.OrderedDict()
        for p in self.parameters:
            out[f"{path}.{p.name}"] = p
        for c in self.children:
            out.update(c.named_parameters(path))
        return out

    def _param(self, shape, name):
        p = th.shared(numpy.zeros(shape), name=name)
        self.parameters.append(p)
        return p


class Initializable(Brick):
    def __init__(self, weights_init=None, biases_init=None, use_bias=True, seed=None, **kwargs):
        super().__init__(**kwargs)
        self.weights_init, self.biases_init, self.use_bias = weights_init, biases_init, use_bias


class Random(Brick):
    @property
    def theano_rng(self):
        return th._RandomStreams(1)


class Linear(Initializable):
    def __init__(self, input_dim=None, output_dim=None, **kwargs):
        super().__init__(**kwargs)
        self.input_dim, self.output_dim = input_dim, output_dim

    def _allocate(self):
        self.W = self._param((self.input_dim, self.output_dim), 'W')
        self.b = self._param((self.output_dim,), 'b')

    @application(inputs=['input_'], outputs=['output'])
    def apply(self, input_):
        return th.dot(input_, self.W) + self.b


class Fork(Initializable):
    def __init__(self, output_names, input_dim=None, output_dims=None, prototype=None, **kwargs):
        super().__init__(**kwargs)
        self.output_names, self.input_dim, self.output_dims = list(output_names), input_dim, output_dims
        self.children = [Linear(name='fork_' + n) for n in self.output_names]

    def _push_allocation_config(self):
        for child, od in zip(self.children, self.output_dims):
            child.input_dim, child.output_dim = self.input_dim, od

    @application(inputs=['input_'])
    def apply(self, input_):
        return [c.apply(input_) for c in self.children]

    @apply.property('outputs')
    def apply_outputs(self):
        return self.output_names


class LookupTable(Initializable):
    def __init__(self, length=None, dim=None, **kwargs):
        super().__init__(**kwargs)
        self.length, self.dim = length, dim

    def _allocate(self):
        self.W = self._param((self.length, self.dim), 'W')

    @application(inputs=['indices'], outputs=['output'])
    def apply(self, indices):
        return self.W[indices.flatten()].reshape(tuple(indices.shape) + (self.dim,))


class GatedRecurrent(Initializable):
    def __init__(self, dim=None, activation=None, gate_activation=None, **kwargs):
        super().__init__(**kwargs)
        self.dim = dim

    def get_dim(self, name):
        if name in ('inputs', 'states'):
            return self.dim
        if name == 'gate_inputs':
            return 2 * self.dim
        if name == 'mask':
            return 0
        raise ValueError(name)

    def _allocate(self):
        self.state_to_state = self._param((self.dim, self.dim), 'state_to_state')
        self.state_to_gates = self._param((self.dim, 2 * self.dim), 'state_to_gates')
        self.initial_state_ = self._param((self.dim,), 'initial_state')

    def _step(self, inputs, gate_inputs, states, mask=None):
        gate_values = th.sigmoid(th.dot(states, self.state_to_gates) + gate_inputs)
        update_values = gate_values[:, :self.dim]
        reset_values = gate_values[:, self.dim:]
        states_reset = states * reset_values
        next_states = th.tanh(th.dot(states_reset, self.state_to_state) + inputs)
        next_states = next_states * update_values + states * (1 - update_values)
        if mask is not None:
            next_states = mask[:, None] * next_states + (1 - mask[:, None]) * states
        return next_states

    @application(sequences=['mask', 'inputs', 'gate_inputs'], states=['states'], outputs=['states'], contexts=[])
    def apply(self, inputs=None, gate_inputs=None, states=None, mask=None, iterate=True, reverse=False,
              return_initial_states=False):
        if not iterate:
            return self._step(inputs, gate_inputs, states, mask)
        n, batch = inputs.shape[0], inputs.shape[1]
        h = self.initial_states(batch) if states is None else states
        outs = []
        for i in (range(n - 1, -1, -1) if reverse else range(n)):
            h = self._step(inputs[i], gate_inputs[i], h, None if mask is None else mask[i])
            outs.append(h)
        return th.stack(outs, 0)

    @application(outputs=['states'])
    def initial_states(self, batch_size, *args, **kwargs):
        return th.repeat(self.initial_state_[None, :], batch_size, 0)


class Bidirectional(Initializable):
    def __init__(self, prototype, **kwargs):
        super().__init__(**kwargs)
        self.prototype = prototype
        self.children = [copy.deepcopy(prototype) for _ in range(2)]
        self.children[0].name, self.children[1].name = 'forward', 'backward'

    @application
    def apply(self, *args, **kwargs):
        forward = self.children[0].apply(*args, as_list=True, **kwargs)
        backward = [x[::-1] for x in self.children[1].apply(*args, reverse=True, as_list=True, **kwargs)]
        return [th.concatenate([f, b], axis=2) for f, b in zip(forward, backward)]


# ------------------------------------------------------------------ blocks.utils / blocks.roles
def shared_floatx_zeros(shape, name=None, **kwargs):
    return th.shared(numpy.zeros(shape), name=name)


def shared_floatx(value, name=None, **kwargs):
    return th.shared(numpy.asarray(value), name=name)


def dict_union(*dicts, **kwargs):
    out = collections.OrderedDict()
    for d in dicts:
        for k, v in d.items():
            if k in out:
                raise ValueError(f"keys overlap: {k}")
            out[k] = v
    return out


def add_role(var, role):
    roles = getattr(var, 'roles', [])
    roles.append(role)
    try:
        var.roles = roles
    except AttributeError:
        pass


def build_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m
    base = mod('blocks.bricks.base', lazy=lazy, application=application, Brick=Brick)
    lookup = mod('blocks.bricks.lookup', LookupTable=LookupTable)
    parallel = mod('blocks.bricks.parallel', Fork=Fork)
    recurrent = mod('blocks.bricks.recurrent', GatedRecurrent=GatedRecurrent, Bidirectional=Bidirectional)
    bricks = mod('blocks.bricks', Initializable=Initializable, Linear=Linear, Random=Random, Brick=Brick, base=base,
                 lookup=lookup, parallel=parallel, recurrent=recurrent)
    roles = mod('blocks.roles', add_role=add_role, INITIAL_STATE='INITIAL_STATE', PARAMETER='PARAMETER',
                WEIGHT='WEIGHT', BIAS='BIAS')
    utils = mod('blocks.utils', shared_floatx_zeros=shared_floatx_zeros, shared_floatx=shared_floatx,
                dict_union=dict_union)
    blocks = mod('blocks', bricks=bricks, roles=roles, utils=utils)
    return {m.__name__: m for m in (blocks, bricks, base, lookup, parallel, recurrent, roles, utils)}
