"""Operator library of the SampleRNN code base on the HIP kernels -- mirrors the on-path functions of
reference sampleRNN/lib/ops.py: ``Linear`` (:32-128, weight norm :101-110), ``Embedding`` (:252-266),
``softmax_and_sample`` / ``softmax_and_argmax`` (:268-297), ``LowMemGRU`` (:395-440) and ``stackedGRU``
(:612-777).  Same names, argument order and parameter names (global registry, parrot_amd.sampleRNN.lib).

Dense products run in libparrot_hip.so (parrot_amd.ops.linear / gru_seq, differentiable); the weight
norm scaling, the embedding gather and the softmax are small torch tensor ops around them.
Off-path functions of the reference file (Batchnorm, MLP, GMM, conv1d, T_one_hot, LSTM variants) are
not provided: nothing on the Char2Wav path calls them.  ``LowMemLSTM`` /
``stackedLSTM`` (:461-610, :823-989) run on the HIP LSTM scan.
"""
from __future__ import annotations

import math

import numpy
import torch

from ... import ops as hip
from .. import lib

_rng = numpy.random.RandomState(234)  # the reference seeds its sampler with 234 (ops.py:15-16)
_gen = None


def uniform(stdev, size):
    """ops.py:19-30."""
    return _rng.uniform(low=-stdev * numpy.sqrt(3), high=stdev * numpy.sqrt(3), size=size).astype('float32')


def _initial_weight(initialization, inp_dim, output_dim):
    if isinstance(initialization, numpy.ndarray):
        assert initialization.shape == (inp_dim, output_dim)
        return initialization
    if initialization == 'lecun' or (initialization is None and inp_dim != output_dim):
        return uniform(numpy.sqrt(1. / inp_dim), (inp_dim, output_dim))
    if initialization == 'glorot':
        return uniform(numpy.sqrt(2. / (inp_dim + output_dim)), (inp_dim, output_dim))
    if initialization == 'he':
        return uniform(numpy.sqrt(2. / inp_dim), (inp_dim, output_dim))
    if initialization == 'glorot_he':
        return uniform(numpy.sqrt(4. / (inp_dim + output_dim)), (inp_dim, output_dim))
    if initialization == 'orthogonal' or (initialization is None and inp_dim == output_dim):
        a = _rng.normal(0.0, 1.0, (inp_dim, output_dim))
        u, _, v = numpy.linalg.svd(a, full_matrices=False)
        q = u if u.shape == (inp_dim, output_dim) else v
        return q.astype('float32')
    raise Exception("Invalid initialization ({})!".format(repr(initialization)))


def effective_weight(name, i, weightnorm=True):
    """W_eff = W * (g / ||W||_2 per output column) (ops.py:101-110)."""
    w = lib.param(name + '.W' + str(i))
    if not weightnorm:
        return w
    g = lib.param(name + '.g' + str(i))
    if w.is_cuda:  # one HIP pass each way (include/parrot_hip.h: samplernn_weightnorm_fold)
        return hip.weightnorm_fold(w, g)
    return w * (g / w.norm(2, dim=0)).unsqueeze(0)


def Linear(name, input_dims, output_dim, inputs, biases=True, initialization=None, weightnorm=True,
           just_params=False):
    """ops.py:32-128: sum_i inputs_i . W_eff_i (+ b); parameters `<name>.W{i}`, `.g{i}`, `.b`."""
    if not isinstance(input_dims, list):
        input_dims, inputs = [input_dims], [inputs]
    params, terms = [], []
    for i, (inp, inp_dim) in enumerate(zip(inputs, input_dims)):
        if (name + '.W' + str(i)) not in lib._params:
            wv = _initial_weight(initialization, inp_dim, output_dim)
            lib.param(name + '.W' + str(i), wv)
            if weightnorm:
                lib.param(name + '.g' + str(i), numpy.linalg.norm(wv, axis=0))
        params.append(lib.param(name + '.W' + str(i)))
        if weightnorm:
            params.append(lib.param(name + '.g' + str(i)))
        if not just_params:
            terms.append((inp, effective_weight(name, i, weightnorm)))
    b = None
    if biases:
        b = lib.param(name + '.b', numpy.zeros((output_dim,), dtype='float32'))
        params.append(b)
    if just_params:
        return params
    out = None
    for k, (inp, w) in enumerate(terms):
        y = hip.linear(inp.to(torch.float32), w, b if k == 0 else None)
        out = y if out is None else out + y
    return out


def Embedding(name, n_symbols, output_dim, indices):
    """ops.py:252-266: vectors[indices]."""
    vectors = lib.param(name, _rng.randn(n_symbols, output_dim).astype('float32')) \
        if name not in lib._params else lib.param(name)
    if indices is None:  # (extension) the table itself: callers that fold the gather into a fused operator
        return vectors
    return vectors[indices.reshape(-1).long()].reshape(*indices.shape, output_dim)


def softmax_and_sample(logits, temperature=1.):
    """ops.py:268-294.  temperature 0 -> argmax of the softmax; > 0 -> multinomial draw (torch's
    generator seeded with 234; Theano's MRG stream itself is not reproducible here)."""
    global _gen
    assert temperature >= 0, "`temperature` should be a non-negative value!"
    flat = logits.reshape(-1, logits.shape[-1])
    if temperature == 0:
        out = torch.softmax(flat, -1).argmax(-1)
    else:
        if _gen is None or _gen.device != flat.device:
            _gen = torch.Generator(device=flat.device).manual_seed(234)
        out = torch.multinomial(torch.softmax(flat / temperature, -1), 1, generator=_gen)[:, 0]
    return out.reshape(logits.shape[:-1])


def softmax_and_argmax(logits):
    """ops.py:296-297."""
    return softmax_and_sample(logits, temperature=0)


def LowMemGRU(name, input_dim, hidden_dim, inputs, h0=None, mask=None, weightnorm=True):
    """ops.py:395-440 with the step of :329-393: inputs [B,n,input_dim] -> states [B,n,hidden_dim].
    The Input linear (with bias) is applied to all steps at once (same arithmetic as inside the step),
    the recurrence runs in the HIP GRU scan."""
    step = name + '.Step'
    processed = Linear(step + '.Input', input_dim, 3 * hidden_dim, inputs, weightnorm=weightnorm)  # [B,n,3H]
    Linear(step + '.Recurrent_Gates', hidden_dim, 2 * hidden_dim, None, biases=False, weightnorm=weightnorm,
           just_params=True)
    Linear(step + '.Recurrent_Candidate', hidden_dim, hidden_dim, None, biases=False, initialization='orthogonal',
           weightnorm=weightnorm, just_params=True)
    Wg = effective_weight(step + '.Recurrent_Gates', 0, weightnorm).contiguous()
    Wc = effective_weight(step + '.Recurrent_Candidate', 0, weightnorm).contiguous()
    pt = processed.transpose(0, 1)  # time-major
    gate_in = pt[..., :2 * hidden_dim].contiguous()
    cand_in = pt[..., 2 * hidden_dim:].contiguous()
    if h0 is None:
        h0v = lib.param(name + '.Recurrent.h0_0', numpy.zeros((hidden_dim,), dtype='float32'))
        h0 = h0v.unsqueeze(0).expand(inputs.shape[0], -1)
    out = hip.gru_seq(cand_in, gate_in, h0.contiguous(), Wc, Wg)
    return out.transpose(0, 1)


def stackedGRU(name, n_rnn, input_dim, hidden_dim, inputs, h0, weightnorm, skip_conn):
    """ops.py:612-777; h0 [B, n_rnn, hidden_dim].  Returns (out [B,n,hidden], last_hiddens [B,n_rnn,hidden]).
    skip_conn (ops.py:650-695): layer k > 1 reads [h_{k-1} ; inputs] and the output is the sum of one Linear per layer."""
    assert n_rnn in range(1, 6), "n_rnn should be in [1,2,3,4,5]"
    assert not (n_rnn == 1 and skip_conn), "Single layer RNN cannot have skip connections"
    prev, out = inputs, None
    last = []
    for layer in range(n_rnn):
        k = layer + 1
        if layer == 0:
            inp, dim, nm = inputs, input_dim, name + '1'
        elif not skip_conn:
            inp, dim, nm = prev, hidden_dim, name + str(k)
        else:
            inp, dim, nm = torch.cat([prev, inputs], dim=-1), hidden_dim + input_dim, name + str(k) + '+inpskip'
        prev = LowMemGRU(nm, dim, hidden_dim, inp, h0=h0[:, layer], weightnorm=weightnorm)
        last.append(prev[:, -1])
        if skip_conn:  # among the output skips only the first carries a bias (ops.py:655-661)
            y = Linear(name + '.outskip%dy' % k, hidden_dim, hidden_dim, prev, biases=(k == 1), initialization='he',
                       weightnorm=weightnorm)
            out = y if out is None else out + y
        else:
            out = prev
    return out, torch.stack(last, dim=1)


def LowMemLSTM(name, input_dim, hidden_dim, inputs, h0=None, mask=None, weightnorm=True):
    """ops.py:555-610 with the step of :461-553: inputs [B,n,input_dim], h0 [B, 2*hidden] = [s | c] ->
    states [B,n,2*hidden] (s and c concatenated, like the reference).  Gate order i | f | o | g, forget
    bias initialised to 3 (ops.py:469, 524-530)."""
    step = name + '.Step'
    processed = Linear(step + '.Input', input_dim, 4 * hidden_dim, inputs, biases=False, weightnorm=weightnorm)
    Linear(step + '.Recurrent_Gates', hidden_dim, 4 * hidden_dim, None, biases=False, weightnorm=weightnorm,
           just_params=True)
    bias_init = numpy.zeros((4 * hidden_dim,), dtype='float32')
    bias_init[hidden_dim:2 * hidden_dim] = 3.
    b = lib.param(step + '.b', bias_init)
    W = effective_weight(step + '.Recurrent_Gates', 0, weightnorm).contiguous()
    pre_in = (processed + b).transpose(0, 1).contiguous()  # time-major [n,B,4H]
    if h0 is None:
        h0v = lib.param(name + '.Recurrent.h0_0', numpy.zeros((2 * hidden_dim,), dtype='float32'))
        h0 = h0v.unsqueeze(0).expand(inputs.shape[0], -1)
    s0, c0 = h0[:, :hidden_dim], h0[:, hidden_dim:]
    s, c = hip.lstm_seq(pre_in, s0, c0, W)
    return torch.cat([s, c], dim=-1).transpose(0, 1)


def stackedLSTM(name, n_rnn, input_dim, hidden_dim, inputs, h0, weightnorm, skip_conn):
    """ops.py:823-989; h0 [B, n_rnn, 2*hidden_dim].  Returns (out [B,n,hidden], last_hiddens [B,n_rnn,2*hidden]).
    skip_conn (ops.py:861-880) as in stackedGRU; the output skips are named '<name>k.outskipky' here."""
    assert n_rnn in range(1, 6), "n_rnn should be in [1,2,3,4,5]"
    assert not (n_rnn == 1 and skip_conn), "Single layer RNN cannot have skip connections"
    prev, out = inputs, None
    last = []
    for layer in range(n_rnn):
        k = layer + 1
        if layer == 0:
            inp, dim, nm = inputs, input_dim, name + '1'
        elif not skip_conn:
            inp, dim, nm = prev, hidden_dim, name + str(k)
        else:
            inp, dim, nm = torch.cat([prev, inputs], dim=-1), hidden_dim + input_dim, name + str(k) + '+inpskip'
        full = LowMemLSTM(nm, dim, hidden_dim, inp, h0=h0[:, layer], weightnorm=weightnorm)
        last.append(full[:, -1])
        prev = full[:, :, :hidden_dim]
        if skip_conn:
            y = Linear(name + '%d.outskip%dy' % (k, k), hidden_dim, hidden_dim, prev, biases=(k == 1),
                       initialization='he', weightnorm=weightnorm)
            out = y if out is None else out + y
        else:
            out = prev
    return out, torch.stack(last, dim=1)
