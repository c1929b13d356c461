"""torch-CPU restatement of the conditional three-tier SampleRNN (test infrastructure, see
oracle/__init__.py).  PINNED (round 2) to vectors produced by executing the reference's own ops.py / three_tier.py on
an eager Theano stand-in (oracle/refshim, tests/golden/make_ref_golden.py; outputs and every gradient at 1e-10).

Restates sampleRNN/lib/ops.py (Linear with weight norm :32-128, Embedding :252-266, GRU step
:329-393, stackedGRU :612-777 for n_rnn = 1, softmax_and_argmax :268-297) and
sampleRNN/models/conditional/three_tier.py (constants :145-202, big_frame_level_rnn :291-380,
frame_level_rnn :382-450, sample_level_predictor :452-515, compute_cost :534-636, the generation loop
:750-851 with the deterministic temperature-0 sampler).

Parameters: dict name -> tensor with the reference's registry names (lib/__init__.py:28-47), e.g.
``BigFrameLevel.GRU1.Step.Input.W0``, ``FrameLevel.h0``, ``SampleLevel.Embedding``.
"""
from __future__ import annotations

import math

import torch

# three_tier.py:145-202 (hard-coded run configuration)
DEFAULTS = dict(BIG_FRAME_SIZE=80, FRAME_SIZE=10, EMB_SIZE=256, DIM=1024, N_RNN=1, Q_LEVELS=256,
                WEIGHT_NORM=True, LEARN_H0=True, SKIP_CONN=False, FEAT_DIM=63, RNN_TYPE='GRU')


def config(**kw):
    c = dict(DEFAULTS)
    c.update(kw)
    c['BIG_DIM'] = c['DIM']
    c['H0_MULT'] = 2 if c['RNN_TYPE'] == 'LSTM' else 1  # three_tier.py:163
    return c


def _uniform(g, stdev, shape):
    """ops.py:19-30."""
    a = stdev * math.sqrt(3.0)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * a


def _orthogonal(g, shape):
    a = torch.randn(shape, generator=g, dtype=torch.float64)
    u, _, vh = torch.linalg.svd(a, full_matrices=False)
    return u if u.shape == tuple(shape) else vh


def _linear_params(p, g, name, input_dims, output_dim, biases=True, init=None, weightnorm=True):
    """Parameter creation of lib.ops.Linear (ops.py:62-124)."""
    if not isinstance(input_dims, (list, tuple)):
        input_dims = [input_dims]
    for i, d in enumerate(input_dims):
        if init == 'he':
            w = _uniform(g, math.sqrt(2.0 / d), (d, output_dim))
        elif init == 'orthogonal' or (init is None and d == output_dim):
            w = _orthogonal(g, (d, output_dim))
        else:  # lecun
            w = _uniform(g, math.sqrt(1.0 / d), (d, output_dim))
        p[f'{name}.W{i}'] = w
        if weightnorm:
            p[f'{name}.g{i}'] = w.norm(dim=0)  # ops.py:101-107: g initialised to the column norms
    if biases:
        p[f'{name}.b'] = torch.zeros(output_dim, dtype=torch.float64)


def init_params(c, seed=1234, dtype=torch.float64, perturb=0.0):
    """All parameters the generation / cost graphs register (three_tier.py:291-515).  perturb > 0 adds
    noise to biases / g / h0 so that tests exercise every term."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    D, BD, wn = c['DIM'], c['BIG_DIM'], c['WEIGHT_NORM']
    BFS, FS, Q, EMB = c['BIG_FRAME_SIZE'], c['FRAME_SIZE'], c['Q_LEVELS'], c['EMB_SIZE']
    _linear_params(p, g, 'BigFrameLevel.rnn_inp_fusion', [BFS, c['FEAT_DIM']], BD, init='he', weightnorm=wn)
    HM = c['H0_MULT']
    p['BigFrameLevel.h0'] = torch.zeros(c['N_RNN'], HM * BD, dtype=torch.float64)
    skip = c['SKIP_CONN']
    for tier, dim in (('BigFrameLevel', BD), ('FrameLevel', D)):
        for layer in range(1, c['N_RNN'] + 1):
            # with skip connections the layers above the first also see the stack's input (ops.py:663-680, 872-888)
            in_dim = dim if (layer == 1 or not skip) else 2 * dim
            tag = '+inpskip' if (skip and layer > 1) else ''
            if skip:  # output skips: Linear(h_layer) summed into the stack's output; only the first carries a bias
                nm = f'{tier}.GRU.outskip{layer}y' if c['RNN_TYPE'] == 'GRU' else f'{tier}.LSTM{layer}.outskip{layer}y'
                _linear_params(p, g, nm, dim, dim, biases=(layer == 1), init='he', weightnorm=wn)
            if c['RNN_TYPE'] == 'GRU':
                pre = f'{tier}.GRU{layer}{tag}.Step'
                _linear_params(p, g, f'{pre}.Input', in_dim, 3 * dim, weightnorm=wn)
                _linear_params(p, g, f'{pre}.Recurrent_Gates', dim, 2 * dim, biases=False, weightnorm=wn)
                _linear_params(p, g, f'{pre}.Recurrent_Candidate', dim, dim, biases=False, init='orthogonal',
                               weightnorm=wn)
            else:  # ops.py:505-530
                pre = f'{tier}.LSTM{layer}{tag}.Step'
                _linear_params(p, g, f'{pre}.Input', in_dim, 4 * dim, biases=False, weightnorm=wn)
                _linear_params(p, g, f'{pre}.Recurrent_Gates', dim, 4 * dim, biases=False, weightnorm=wn)
                b = torch.zeros(4 * dim, dtype=torch.float64)
                b[dim:2 * dim] = 3.0
                p[f'{pre}.b'] = b
    _linear_params(p, g, 'BigFrameLevel.Output', BD, D * BFS // FS, init='he', weightnorm=wn)
    _linear_params(p, g, 'BigFrameLevel.IndependentPreds', BD, Q * BFS, init='he', weightnorm=wn)
    _linear_params(p, g, 'FrameLevel.InputExpand', FS, D, init='he', weightnorm=wn)
    p['FrameLevel.h0'] = torch.zeros(c['N_RNN'], HM * D, dtype=torch.float64)
    _linear_params(p, g, 'FrameLevel.Output', D, FS * D, init='he', weightnorm=wn)
    p['SampleLevel.Embedding'] = torch.randn(Q, EMB, generator=g, dtype=torch.float64)
    _linear_params(p, g, 'SampleLevel.L1_PrevSamples', FS * EMB, D, biases=False, init='he', weightnorm=wn)
    _linear_params(p, g, 'SampleLevel.L2', D, D, init='he', weightnorm=wn)
    _linear_params(p, g, 'SampleLevel.L3', D, D, init='he', weightnorm=wn)
    _linear_params(p, g, 'SampleLevel.Output', D, Q, weightnorm=wn)
    if perturb > 0:
        for k in p:
            if k.endswith('.b') or k.endswith('.h0') or '.g' in k.split('.')[-1]:
                p[k] = p[k] + perturb * torch.randn(p[k].shape, generator=g, dtype=torch.float64) * \
                    (p[k].abs().mean() + 1.0)
    return {k: v.to(dtype) for k, v in p.items()}


def linear(p, c, name, inputs, n_inputs=1, biases=True):
    """lib.ops.Linear forward (ops.py:99-128): sum_i x_i . (W_i * g_i / ||W_i||_col) + b."""
    if n_inputs == 1:
        inputs = [inputs]
    out = 0
    for i, x in enumerate(inputs):
        w = p[f'{name}.W{i}']
        if c['WEIGHT_NORM']:
            w = w * (p[f'{name}.g{i}'] / w.norm(dim=0))[None, :]
        out = out + x @ w
    if biases:
        out = out + p[f'{name}.b']
    return out


def gru_step(p, c, name, dim, x, h):
    """__GRUStep (ops.py:329-393)."""
    pi = linear(p, c, f'{name}.Input', x)
    gates = torch.sigmoid(linear(p, c, f'{name}.Recurrent_Gates', h, biases=False) + pi[:, :2 * dim])
    update, reset = gates[:, :dim], gates[:, dim:]
    cand = torch.tanh(linear(p, c, f'{name}.Recurrent_Candidate', reset * h, biases=False) + pi[:, 2 * dim:])
    return update * cand + (1 - update) * h


def lstm_step(p, c, name, dim, x, hc):
    """__LSTMStep (ops.py:461-553): hc = [s | c]; gate order i | f | o | g."""
    s_tm1, c_tm1 = hc[:, :dim], hc[:, dim:]
    pre = linear(p, c, f'{name}.Input', x, biases=False) + linear(p, c, f'{name}.Recurrent_Gates', s_tm1, biases=False) \
        + p[f'{name}.b']
    gates = torch.sigmoid(pre[:, :3 * dim])
    i, f, o = gates[:, :dim], gates[:, dim:2 * dim], gates[:, 2 * dim:]
    g = torch.tanh(pre[:, 3 * dim:])
    c_t = c_tm1 * f + g * i
    return torch.cat([torch.tanh(c_t) * o, c_t], -1)


def stacked_gru(p, c, name, dim, inputs, h0):
    """stackedGRU / stackedLSTM (ops.py:612-777, 823-989): inputs [B,n,dim], h0 [B,n_rnn,H0_MULT*dim].  `name` ends in
    '.GRU'; the LSTM parameters use '.LSTM' instead.  With SKIP_CONN (Graves' stacks, ops.py:650-695, 861-880) layer k > 1
    reads [h_{k-1} ; inputs] (its parameters are named '<base>k+inpskip') and the stack's output is the sum of one Linear
    per layer ('<GRU base>.outskipky' / '<LSTM base>k.outskipky'; only the first has a bias)."""
    lstm = c['RNN_TYPE'] == 'LSTM'
    skip = c['SKIP_CONN']
    assert not (skip and c['N_RNN'] == 1), "Single layer RNN cannot have skip connections"
    base = name[:-4] + ('.LSTM' if lstm else '.GRU')
    x = inputs
    lasts = []
    out = None
    for layer in range(c['N_RNN']):
        k = layer + 1
        h = h0[:, layer]
        tag = '+inpskip' if (skip and k > 1) else ''
        xin = torch.cat([x, inputs], -1) if (skip and k > 1) else x
        outs = []
        for t in range(xin.shape[1]):
            h = (lstm_step if lstm else gru_step)(p, c, f'{base}{k}{tag}.Step', dim, xin[:, t], h)
            outs.append(h)
        full = torch.stack(outs, 1)
        lasts.append(full[:, -1])
        x = full[:, :, :dim]
        if skip:
            nm = f'{base}{k}.outskip{k}y' if lstm else f'{base}.outskip{k}y'
            y = linear(p, c, nm, x, biases=(k == 1))
            out = y if out is None else out + y
    return (out if skip else x), torch.stack(lasts, 1)


def _frames_to_float(frames, c):
    """three_tier.py:309-310, 398-399: (s / (Q/2) - 1) * 2."""
    dt = torch.float64
    return ((frames.to(dt) / (c['Q_LEVELS'] // 2)) - 1.0) * 2.0


def big_frame_level_rnn(p, c, input_sequences, h0, reset, features):
    """three_tier.py:291-380."""
    B = input_sequences.shape[0]
    BFS, FS, D, BD, Q = c['BIG_FRAME_SIZE'], c['FRAME_SIZE'], c['DIM'], c['BIG_DIM'], c['Q_LEVELS']
    frames = _frames_to_float(input_sequences.reshape(B, -1, BFS), c).to(p['BigFrameLevel.h0'].dtype)
    rnn_inp = linear(p, c, 'BigFrameLevel.rnn_inp_fusion', [frames, features.to(frames.dtype)], n_inputs=2)
    if reset:
        h0 = p['BigFrameLevel.h0'][None].expand(B, -1, -1)
    rnns_out, last_hidden = stacked_gru(p, c, 'BigFrameLevel.GRU', BD, rnn_inp, h0)
    output = linear(p, c, 'BigFrameLevel.Output', rnns_out)
    output = output.reshape(B, output.shape[1] * BFS // FS, D)
    indep = linear(p, c, 'BigFrameLevel.IndependentPreds', rnns_out)
    indep = indep.reshape(B, indep.shape[1] * BFS, Q)
    return output, last_hidden, indep


def frame_level_rnn(p, c, input_sequences, other_input, h0, reset):
    """three_tier.py:382-450."""
    B = input_sequences.shape[0]
    FS, D = c['FRAME_SIZE'], c['DIM']
    frames = _frames_to_float(input_sequences.reshape(B, -1, FS), c).to(p['FrameLevel.h0'].dtype)
    gru_input = linear(p, c, 'FrameLevel.InputExpand', frames) + other_input
    if reset:
        h0 = p['FrameLevel.h0'][None].expand(B, -1, -1)
    rnns_out, last_hidden = stacked_gru(p, c, 'FrameLevel.GRU', D, gru_input, h0)
    output = linear(p, c, 'FrameLevel.Output', rnns_out)
    output = output.reshape(B, output.shape[1] * FS, D)
    return output, last_hidden


def _relu_at_ties(pre, tie_mask, tie_tol, report):
    """relu(pre), except that an element within tie_tol * max|pre| of the kink takes the branch `tie_mask` names
    (the implementation under test's own mask): at pre = 0 +- rounding the branch -- and one whole term of every upstream
    gradient -- is decided by the last bit of an f32 product, so a checker that insists on its own float64 sign there
    tests luck, not arithmetic (the same policy as the arg-max ties of the greedy generation check)."""
    own = pre > 0
    if tie_mask is None:
        return torch.relu(pre)
    near = pre.detach().abs() < tie_tol * pre.detach().abs().max()
    use = torch.where(near, tie_mask.to(own.device), own)
    if report is not None:
        report.append((int(near.sum()), int((use != own).sum())))
    return pre * use.to(pre.dtype)


def sample_level_predictor(p, c, frame_level_outputs, prev_samples, relu_ties=None, tie_tol=1e-6, tie_report=None):
    """three_tier.py:452-515.  relu_ties: optional [mask of L2, mask of L3] (bool, the tested implementation's relu
    outputs > 0), consulted only for pre-activations within tie_tol of 0 (see _relu_at_ties)."""
    FS, EMB = c['FRAME_SIZE'], c['EMB_SIZE']
    emb = p['SampleLevel.Embedding'][prev_samples.reshape(-1).long()].reshape(-1, FS * EMB)
    out = linear(p, c, 'SampleLevel.L1_PrevSamples', emb, biases=False) + frame_level_outputs
    out = _relu_at_ties(linear(p, c, 'SampleLevel.L2', out), relu_ties and relu_ties[0], tie_tol, tie_report)
    out = _relu_at_ties(linear(p, c, 'SampleLevel.L3', out), relu_ties and relu_ties[1], tie_tol, tie_report)
    return linear(p, c, 'SampleLevel.Output', out)


def compute_cost(p, c, sequences, features, h0, big_h0, reset, mask, relu_ties=None, tie_tol=1e-6, tie_report=None):
    """three_tier.py:534-636.  sequences [B, S+80] int, features [B, S/80, 63], mask [B, S+80].
    Returns (cost_bits, ip_cost_bits, new_h0, new_big_h0).  relu_ties / tie_tol / tie_report: see sample_level_predictor."""
    BFS, FS, D, Q = c['BIG_FRAME_SIZE'], c['FRAME_SIZE'], c['DIM'], c['Q_LEVELS']
    big_in = sequences[:, :-BFS]
    inp = sequences[:, BFS - FS:-FS]
    target = sequences[:, BFS:]
    tmask = mask[:, BFS:]
    big_out, new_big_h0, indep = big_frame_level_rnn(p, c, big_in, big_h0, reset, features)
    frame_out, new_h0 = frame_level_rnn(p, c, inp, big_out, h0, reset)
    prev = sequences[:, BFS - FS:-1]
    prev = prev.unfold(1, FS, 1).reshape(-1, FS)  # images2neibs, stride 1 (three_tier.py:555-558)
    logits = sample_level_predictor(p, c, frame_out.reshape(-1, D), prev, relu_ties, tie_tol, tie_report)
    lse = torch.logsumexp(logits, -1)
    ce = (lse - logits.gather(1, target.reshape(-1, 1).long())[:, 0]).reshape(target.shape)
    log2e = math.log2(math.e)
    cost = (ce * tmask).sum() / (tmask.sum() + 1e-5) * log2e
    il = indep.reshape(-1, Q)
    ice = (torch.logsumexp(il, -1) - il.gather(1, target.reshape(-1, 1).long())[:, 0]).reshape(target.shape)
    ip_cost = (ice * tmask).sum() / (tmask.sum() + 1e-5) * log2e
    return cost, ip_cost, new_h0, new_big_h0


def generate(p, c, features, return_logits=False):
    """generate_and_save_samples loop (three_tier.py:794-832) with the temperature-0 sampler
    (softmax_and_argmax, ops.py:296-297; argmax ties -> lowest index).  features [T,B,63] time-major.
    Returns samples [B, 80*T] int32 (first 80 = Q_ZERO)."""
    BFS, FS, D = c['BIG_FRAME_SIZE'], c['FRAME_SIZE'], c['DIM']
    feats = features.transpose(0, 1)
    B, LENGTH = feats.shape[0], feats.shape[1] * BFS
    samples = torch.zeros(B, LENGTH, dtype=torch.int64)
    samples[:, :BFS] = c['Q_LEVELS'] // 2
    dt = p['FrameLevel.h0'].dtype
    big_h0 = torch.zeros(B, c['N_RNN'], c['H0_MULT'] * c['BIG_DIM'], dtype=dt)
    h0 = torch.zeros(B, c['N_RNN'], c['H0_MULT'] * D, dtype=dt)
    big_out = frame_out = None
    all_logits = []
    for t in range(BFS, LENGTH):
        if t % BFS == 0:
            big_out, big_h0, _ = big_frame_level_rnn(p, c, samples[:, t - BFS:t], big_h0, t == BFS,
                                                     feats[:, t // BFS][:, None])
        if t % FS == 0:
            frame_out, h0 = frame_level_rnn(p, c, samples[:, t - FS:t],
                                            big_out[:, (t // FS) % (BFS // FS)][:, None], h0, t == BFS)
        logits = sample_level_predictor(p, c, frame_out[:, t % FS], samples[:, t - FS:t])
        if return_logits:
            all_logits.append(logits)
        samples[:, t] = torch.argmax(torch.softmax(logits, -1), -1)
    if return_logits:
        return samples.to(torch.int32), torch.stack(all_logits, 1)
    return samples.to(torch.int32)
