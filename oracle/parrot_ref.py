"""torch-CPU restatement of reference model.py (test infrastructure, see oracle/__init__.py).

PINNED (round 2) to vectors produced by executing the reference's own model.py on eager Theano / Blocks
stand-ins (oracle/refshim, tests/golden/make_ref_golden.py, tests/test_ref_golden_cpu.py: outputs and every gradient
at 1e-10).  Every function cites the reference lines it restates.  Row-vector convention
``y = x . W + b`` with ``W [in, out]`` (Blocks Linear).  Default dtype float64 (the checker);
``dtype=torch.float32`` is used when this code is timed as the CPU baseline.

Parameters live in a flat dict keyed by Blocks-style brick paths (sample.py:83 shows the style:
``/parrot/lookuptable.W``), see ``init_params``.  Generalisation beyond the reference: the number
of decoder layers L may be 1, 2 or 3 (the reference hard-wires 3, model.py:312-314), and
``cell_type='lstm'`` swaps the GatedRecurrent layers for LSTM layers (BASELINE configs[3]; the reference's
own model.py has no LSTM decoder, so the cell algebra is the one the reference does contain,
sampleRNN/lib/ops.py:505-553: gates i|f|o|g, state = [s | c]; each Fork then has the single output
``rnn{l}_inputs`` of width 4H and the layer owns ``W_state`` [H,4H], ``initial_state``, ``initial_cells``).
"""
from __future__ import annotations

import math

import torch

SQRT_INV_2PI = 0.3989422917366028  # model.py:682


# ----------------------------------------------------------------------------- config / params
def default_config(**kw):
    """Parrot.__init__ keyword defaults (model.py:251-277) plus `num_layers` and
    `encoder_literal` (the reference scans the encoder GRU over the BATCH axis)."""
    cfg = dict(
        input_dim=420, output_dim=63, rnn_h_dim=1024, readouts_dim=1024,
        weak_feedback=False, full_feedback=False, feedback_noise_level=None, layer_norm=False,
        use_speaker=False, num_speakers=21, speaker_dim=128, which_cost='MSE', k_gmm=20,
        sampling_bias=0., epsilon=1e-5, num_characters=43, attention_type='graves',
        attention_size=10, attention_alignment=1., sharpening_coeff=1., timing_coeff=1.,
        encoder_type=None, encoder_dim=128, raw_output=False,
        num_layers=3, encoder_literal=True, cell_type='gru', lstm_forget_bias=3.0)
    cfg.update(kw)
    if cfg['full_feedback']:
        cfg['weak_feedback'] = True  # model.py:485
    cfg['encoded_input_dim'] = (2 * cfg['encoder_dim'] if cfg['encoder_type'] == 'bidirectional'
                                else cfg['input_dim'])  # model.py:302-307
    return cfg


def layer_outs(cfg, l):
    """Fork output names / widths feeding layer l (model.py:331-349 for the GRU: inputs H, gates 2H)."""
    H = cfg['rnn_h_dim']
    if cfg.get('cell_type', 'gru') == 'lstm':
        return [(f'rnn{l}_inputs', 4 * H)]
    return [(f'rnn{l}_inputs', H), (f'rnn{l}_gates', 2 * H)]


def _names(cfg, l):
    return [n for n, _ in layer_outs(cfg, l)]


def param_shapes(cfg):
    """name -> shape for every parameter of the configured model (Blocks brick paths)."""
    H, R, O, E = cfg['rnn_h_dim'], cfg['readouts_dim'], cfg['output_dim'], cfg['encoded_input_dim']
    A, L = cfg['attention_size'], cfg['num_layers']
    s = {}

    def fork(name, din, outs):
        for oname, od in outs:
            s[f'/parrot/{name}/fork_{oname}.W'] = (din, od)
            s[f'/parrot/{name}/fork_{oname}.b'] = (od,)

    def linear(name, din, dout):
        s[f'/parrot/{name}.W'] = (din, dout)
        s[f'/parrot/{name}.b'] = (dout,)

    if cfg['encoder_type'] == 'bidirectional':
        D, ED = cfg['input_dim'], cfg['encoder_dim']
        s['/parrot/encoder/embed_label.W'] = (cfg['num_characters'], D)
        for d in ('forward', 'backward'):
            p = f'/parrot/encoder/encoder/{d}'
            s[f'{p}/fork/fork_inputs.W'] = (D, ED)
            s[f'{p}/fork/fork_inputs.b'] = (ED,)
            s[f'{p}/fork/fork_gate_inputs.W'] = (D, 2 * ED)
            s[f'{p}/fork/fork_gate_inputs.b'] = (2 * ED,)
            s[f'{p}/gatedrecurrent.state_to_state'] = (ED, ED)
            s[f'{p}/gatedrecurrent.state_to_gates'] = (ED, 2 * ED)
            s[f'{p}/gatedrecurrent.initial_state'] = (ED,)
    lstm = cfg.get('cell_type', 'gru') == 'lstm'
    for l in range(1, L + 1):
        if lstm:
            s[f'/parrot/rnn{l}.W_state'] = (H, 4 * H)
            s[f'/parrot/rnn{l}.initial_cells'] = (H,)
        else:
            s[f'/parrot/rnn{l}.state_to_state'] = (H, H)
            s[f'/parrot/rnn{l}.state_to_gates'] = (H, 2 * H)
        s[f'/parrot/rnn{l}.initial_state'] = (H,)
        linear(f'h{l}_to_readout', H, R)
        fork(f'inp_to_h{l}', E, layer_outs(cfg, l))
        for j in range(1, l):
            fork(f'h{j}_to_h{l}', H, layer_outs(cfg, l))
    fork('h1_to_att', H, [('alpha', A), ('beta', A), ('kappa', A)])
    linear('att_to_readout', E, R)
    if cfg['which_cost'] == 'MSE':
        linear('readout_to_output', R, O)
    else:
        K = cfg['k_gmm']
        fork('readout_to_output', R, [('gmm_mu', O * K), ('gmm_sigma', O * K), ('gmm_coeff', K)])
    if cfg['use_speaker']:
        SD = cfg['speaker_dim']
        s['/parrot/lookuptable.W'] = (cfg['num_speakers'], SD)
        for l in range(1, L + 1):
            fork(f'speaker_to_h{l}', SD, layer_outs(cfg, l))
        linear('speaker_to_readout', SD, R)
        if cfg['which_cost'] == 'MSE':
            linear('speaker_to_output', SD, O)
        else:
            K = cfg['k_gmm']
            fork('speaker_to_output', SD, [('gmm_mu', O * K), ('gmm_sigma', O * K), ('gmm_coeff', K)])
    if cfg['weak_feedback']:
        fork('out_to_h1', O, layer_outs(cfg, 1))
    if cfg['full_feedback']:
        for l in range(2, L + 1):
            fork(f'out_to_h{l}', O, layer_outs(cfg, l))
    s['/parrot.initial_w'] = (E,)
    return s


def init_params(cfg, seed=1234, std=0.01, dtype=torch.float64, scale_by_fan_in=False):
    """train.py:30-31: weights ~ N(0, 0.01^2), biases 0, initial states 0 (model.py:502-506).
    scale_by_fan_in=True draws N(0, 1/fan_in) instead so gates/attention leave the linear regime
    (SURVEY 8d second parameter set)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(cfg).items():
        is_w = (name.endswith('.W') or name.endswith('state_to_state') or name.endswith('state_to_gates')
                or name.endswith('W_state'))
        if is_w:
            sd = (1.0 / math.sqrt(shape[0])) if scale_by_fan_in else std
            p[name] = (torch.randn(shape, generator=g, dtype=torch.float64) * sd).to(dtype)
        else:
            if scale_by_fan_in:
                p[name] = (torch.randn(shape, generator=g, dtype=torch.float64) * 0.1).to(dtype)
            else:
                p[name] = torch.zeros(shape, dtype=dtype)
    if cfg.get('cell_type', 'gru') == 'lstm':  # forget-gate bias, sampleRNN/lib/ops.py:526 (FORGET_BIAS = 3)
        H = cfg['rnn_h_dim']
        for l in range(1, cfg['num_layers'] + 1):
            p[f'/parrot/inp_to_h{l}/fork_rnn{l}_inputs.b'][H:2 * H] += cfg.get('lstm_forget_bias', 3.0)
    return p


# ----------------------------------------------------------------------------- bricks
def simple_norm(x, eps=1e-5):
    """model.py:24-27: (x - mean) / (eps + std), population std, no affine."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, unbiased=False, keepdim=True)
    return (x - mean) / (eps + std)


def apply_norm(x, layer_norm):
    """model.py:30-34."""
    return simple_norm(x) if layer_norm else x


# Operand rounding of the DECODER's matrix products (None = exact).  'bf16' restates the product's bf16 operand mode
# (BASELINE configs[3]: weights AND activations rounded to bf16, nearest even, where they enter a product of the scan
# steps, the feedback / speaker projections and the readouts; accumulation, states, the attention window, the encoder and
# the cost stay in the working precision) so that the tests can tell what the MODE costs against the exact reference
# from what the kernels add to it.  Set through `operand_rounding(...)`; never set by the reference itself.
_OPERAND_ROUNDING = None


class operand_rounding:
    def __init__(self, mode):
        assert mode in (None, 'bf16')
        self.mode = mode

    def __enter__(self):
        global _OPERAND_ROUNDING
        self.prev, _OPERAND_ROUNDING = _OPERAND_ROUNDING, self.mode
        return self

    def __exit__(self, *exc):
        global _OPERAND_ROUNDING
        _OPERAND_ROUNDING = self.prev
        return False


_ROUNDED_WEIGHTS = {}  # id(W) -> (W, rounded W): a weight is rounded once per compute_cost call, not once per step


def _rnd(x):
    return x if _OPERAND_ROUNDING is None else x.to(torch.bfloat16).to(x.dtype)


def _mm(x, W):
    """x . W with the operand rounding of the active mode (exact by default)."""
    if _OPERAND_ROUNDING is None:
        return x @ W
    hit = _ROUNDED_WEIGHTS.get(id(W))
    if hit is None or hit[0] is not W:
        hit = _ROUNDED_WEIGHTS[id(W)] = (W, _rnd(W))
    return _rnd(x) @ hit[1]


def linear(p, name, x):
    """Blocks Linear.apply: x . W + b."""
    return _mm(x, p[f'/parrot/{name}.W']) + p[f'/parrot/{name}.b']


def fork(p, name, x, outs):
    """Blocks Fork.apply: one independent Linear per output name, in output_names order."""
    mm = (lambda a, b: a @ b) if name == 'h1_to_att' else _mm  # (the attention window stays exact in every operand mode)
    return [mm(x, p[f'/parrot/{name}/fork_{o}.W']) + p[f'/parrot/{name}/fork_{o}.b'] for o in outs]


def gru_step(inputs, gate_inputs, h, W_ss, W_sg, mask=None):
    """Blocks GatedRecurrent.apply(iterate=False) (call sites model.py:659-662); twin algebra
    sampleRNN/lib/ops.py:364-393.  z = update (first half), r = reset (second half)."""
    H = h.shape[-1]
    g = torch.sigmoid(_mm(h, W_sg) + gate_inputs)
    z, r = g[..., :H], g[..., H:]
    c = torch.tanh(_mm(h * r, W_ss) + inputs)
    hn = c * z + h * (1 - z)
    if mask is not None:
        hn = mask[..., None] * hn + (1 - mask[..., None]) * h
    return hn


def lstm_cell(pre_in, s, c, W_state):
    """sampleRNN/lib/ops.py:505-553 (LSTM step, gate order i|f|o|g) with the forget bias kept in the
    learnable bias of `pre_in` (see init_params)."""
    H = s.shape[-1]
    g = _mm(s, W_state) + pre_in
    i, f, o = torch.sigmoid(g[..., :H]), torch.sigmoid(g[..., H:2 * H]), torch.sigmoid(g[..., 2 * H:3 * H])
    gg = torch.tanh(g[..., 3 * H:])
    cn = f * c + i * gg
    return torch.tanh(cn) * o, cn


def cell_step(p, cfg, l, ins, state):
    """One recurrent layer step; `ins` follows layer_outs(cfg, l), `state` is h (GRU) or (s, c) (LSTM).
    Returns (new state, output fed to the layers above / attention / readout)."""
    if cfg.get('cell_type', 'gru') == 'lstm':
        s, c = lstm_cell(ins[0], state[0], state[1], p[f'/parrot/rnn{l}.W_state'])
        return (s, c), s
    h = gru_step(ins[0], ins[1], state, p[f'/parrot/rnn{l}.state_to_state'], p[f'/parrot/rnn{l}.state_to_gates'])
    return h, h


def gru_scan(inputs, gate_inputs, h0, W_ss, W_sg, mask=None):
    """GatedRecurrent.apply over axis 0 (Blocks recurrent bricks iterate axis 0)."""
    hs = []
    h = h0
    for t in range(inputs.shape[0]):
        h = gru_step(inputs[t], gate_inputs[t], h, W_ss, W_sg, None if mask is None else mask[t])
        hs.append(h)
    return torch.stack(hs, 0)


def encoder_apply(p, cfg, labels):
    """Encoder.apply (model.py:233-247): LookupTable -> Bidirectional(RecurrentWithFork(GRU)).
    Literal mode reproduces the reference quirk: the input is batch-major [B,U,D] and Blocks
    recurrents iterate axis 0, so the scan runs over the batch axis with U as the batch
    (model.py:645 passes no mask)."""
    if cfg['encoder_type'] is None:
        return labels  # model.py:235-236
    with operand_rounding(None):  # the encoder is exact in every operand mode
        return _encoder_apply(p, cfg, labels)


def _encoder_apply(p, cfg, labels):
    emb = p['/parrot/encoder/embed_label.W'][labels]  # [B,U,D]
    seq = emb if cfg['encoder_literal'] else emb.transpose(0, 1)
    outs = []
    for d in ('forward', 'backward'):
        pre = f'/parrot/encoder/encoder/{d}'
        x = seq if d == 'forward' else seq.flip(0)
        inp = x @ p[f'{pre}/fork/fork_inputs.W'] + p[f'{pre}/fork/fork_inputs.b']
        gat = x @ p[f'{pre}/fork/fork_gate_inputs.W'] + p[f'{pre}/fork/fork_gate_inputs.b']
        h0 = p[f'{pre}/gatedrecurrent.initial_state'].expand(x.shape[1], -1)
        hs = gru_scan(inp, gat, h0, p[f'{pre}/gatedrecurrent.state_to_state'],
                      p[f'{pre}/gatedrecurrent.state_to_gates'])
        outs.append(hs if d == 'forward' else hs.flip(0))
    out = torch.cat(outs, -1)
    return out if cfg['encoder_literal'] else out.transpose(0, 1)


def attention_step(cfg, a_hat, b_hat, k_hat, k_tm1, ctx, sampling=False):
    """model.py:664-690 (training) / :931-958 (sampling: sharpening, timing)."""
    eps = cfg['epsilon']
    if cfg['attention_type'] == 'softmax':
        a = torch.softmax(a_hat, -1) + eps
    else:
        a = torch.exp(a_hat) + eps
    sharp = cfg['sharpening_coeff'] if sampling else 1.
    timing = cfg['timing_coeff'] if sampling else 1.
    b = torch.exp(b_hat) * sharp + eps
    k = k_tm1 + cfg['attention_alignment'] * torch.exp(k_hat) / timing
    U = ctx.shape[1]
    u = torch.arange(U, dtype=ctx.dtype)[None, None, :]
    a_, b_, k_ = a[..., None], b[..., None], k[..., None]
    if cfg['attention_type'] == 'softmax':
        phi = SQRT_INV_2PI * (a_ * torch.sqrt(b_) * torch.exp(-0.5 * b_ * (k_ - u) ** 2)).sum(1)
    else:
        phi = (a_ * torch.exp(-b_ * (k_ - u) ** 2)).sum(1)
    w = (phi[..., None] * ctx).sum(1)
    return a, k, phi, w


def decoder_step(p, cfg, seq_in, h_tm1, k_tm1, w_tm1, ctx, sampling=False):
    """One `step` of the training scan (model.py:651-724) for L layers.
    seq_in: per layer, the list of additive inputs for this timestep in layer_outs order
    ((cell, gates) for the GRU).  h_tm1: per-layer states (h, or (s, c) for LSTM layers)."""
    L, ln = cfg['num_layers'], cfg['layer_norm']
    states, hs = [], []
    # layer 1: context of the previous step
    ins = fork(p, 'inp_to_h1', w_tm1, _names(cfg, 1))
    st, h1 = cell_step(p, cfg, 1, [a + b for a, b in zip(seq_in[0], ins)], h_tm1[0])
    states.append(st); hs.append(h1)
    a_hat, b_hat, k_hat = fork(p, 'h1_to_att', h1, ['alpha', 'beta', 'kappa'])
    a, k, phi, w = attention_step(cfg, a_hat, b_hat, k_hat, k_tm1, ctx, sampling)
    for l in range(2, L + 1):
        ins = fork(p, f'inp_to_h{l}', w, _names(cfg, l))
        tot = [a_ + b_ for a_, b_ in zip(seq_in[l - 1], ins)]
        for j in range(1, l):
            outs = fork(p, f'h{j}_to_h{l}', hs[j - 1], _names(cfg, l))
            tot = [t_ + apply_norm(o_, ln) for t_, o_ in zip(tot, outs)]
        st, hl = cell_step(p, cfg, l, tot, h_tm1[l - 1])
        states.append(st); hs.append(hl)
    return states, hs, k, w, phi, a


def logsumexp(x, axis):
    """model.py:37-41."""
    x_max = x.max(axis, keepdim=True)[0]
    z = torch.log(torch.exp(x - x_max).sum(axis, keepdim=True)) + x_max
    return z.sum(axis)


def cost_gmm(y, mu, sig, weight):
    """model.py:65-91."""
    shape_y = y.shape
    k = weight.shape[-1]
    y2 = y.reshape(-1, shape_y[-1])[..., None]
    mu = mu.reshape(-1, shape_y[-1], k)
    sig = sig.reshape(-1, shape_y[-1], k)
    weight = weight.reshape(-1, k)
    diff = (y2 - mu) ** 2
    inner = -0.5 * (diff / sig ** 2 + 2 * torch.log(sig) + math.log(2 * math.pi)).sum(-2)
    nll = -logsumexp(torch.log(weight) + inner, -1)
    return nll.reshape(shape_y[:-1])


def initial_carry(p, cfg, batch):
    """Parrot.initial_states (model.py:529-549): learned initial h, learned initial_w, zero k."""
    dt = p['/parrot.initial_w'].dtype
    hs = [p[f'/parrot/rnn{l}.initial_state'].expand(batch, -1) for l in range(1, cfg['num_layers'] + 1)]
    if cfg.get('cell_type', 'gru') == 'lstm':
        hs = [(h, p[f'/parrot/rnn{l + 1}.initial_cells'].expand(batch, -1)) for l, h in enumerate(hs)]
    w = p['/parrot.initial_w'].expand(batch, -1)
    k = torch.zeros(batch, cfg['attention_size'], dtype=dt)
    return dict(h=hs, w=w, k=k)


def compute_cost(p, cfg, features, features_mask, labels, labels_mask, speaker=None, start_flag=1,
                 carry=None, feedback_noise=None):
    """Parrot.compute_cost (model.py:551-824), MSE or GMM-NLL head, without the raw_output branch.

    features [T+1,B,O], features_mask [T+1,B] (time-major), labels [B,U] int64, labels_mask [B,U],
    speaker [B,1] int64 or None.  carry = state left by the previous TBPTT window (used when
    start_flag == 0, model.py:633-643).  Returns (cost, new_carry, attention_vars, extras)."""
    L, H, ln = cfg['num_layers'], cfg['rnn_h_dim'], cfg['layer_norm']
    _ROUNDED_WEIGHTS.clear()
    dt = p['/parrot.initial_w'].dtype
    target = features[1:]
    mask = features_mask[1:]
    T, B = mask.shape
    seq = [[torch.zeros(T, B, wd, dtype=dt) for _, wd in layer_outs(cfg, l)] for l in range(1, L + 1)]

    def add_to(l, outs, first=False):
        seq[l - 1] = [s_ + apply_norm(o_, ln) for s_, o_ in zip(seq[l - 1], outs)]

    if cfg['weak_feedback']:  # model.py:571-588
        inp = features[:-1]
        if feedback_noise is not None:
            inp = inp + feedback_noise
        add_to(1, fork(p, 'out_to_h1', inp, _names(cfg, 1)))
    if cfg['full_feedback']:  # model.py:590-603
        for l in range(2, L + 1):
            add_to(l, fork(p, f'out_to_h{l}', inp, _names(cfg, l)))
    emb_speaker = None
    if cfg['use_speaker']:  # model.py:605-627
        emb_speaker = p['/parrot/lookuptable.W'][speaker[:, 0]][None]
        for l in range(1, L + 1):
            add_to(l, fork(p, f'speaker_to_h{l}', emb_speaker, _names(cfg, l)))

    init = initial_carry(p, cfg, B)
    if start_flag or carry is None:  # model.py:633-643
        h, w, k = init['h'], init['w'], init['k']
    else:
        h, w, k = carry['h'], carry['w'], carry['k']

    ctx = encoder_apply(p, cfg, labels) * labels_mask[..., None]  # model.py:645-646

    hs_all = [[] for _ in range(L)]
    ks, ws, phis, pis = [], [], [], []
    for t in range(T):  # theano.scan, model.py:726-737
        h, outs, k, w, phi, a = decoder_step(p, cfg, [[x[t] for x in s] for s in seq], h, k, w, ctx)
        for l in range(L):
            hs_all[l].append(outs[l])
        ks.append(k); ws.append(w); phis.append(phi); pis.append(a)
    hs_all = [torch.stack(x, 0) for x in hs_all]
    k_all, w_all, phi_all, pi_all = (torch.stack(x, 0) for x in (ks, ws, phis, pis))

    readouts = 0  # model.py:739-753
    for l in range(1, L + 1):
        readouts = readouts + apply_norm(linear(p, f'h{l}_to_readout', hs_all[l - 1]), ln)
    if cfg['use_speaker']:
        readouts = readouts + linear(p, 'speaker_to_readout', emb_speaker)
    readouts = readouts + linear(p, 'att_to_readout', w_all)

    extras = {}
    if cfg['which_cost'] == 'MSE':  # model.py:757-764
        predicted = linear(p, 'readout_to_output', readouts)
        if cfg['use_speaker']:
            predicted = predicted + linear(p, 'speaker_to_output', emb_speaker)
        cost = ((predicted - target) ** 2).sum(-1)
        next_x, coeff = predicted, predicted
    else:  # model.py:765-782 (NLL only; the sampled next_x is stochastic in the reference)
        mu, sigma, coeff = fork(p, 'readout_to_output', readouts, ['gmm_mu', 'gmm_sigma', 'gmm_coeff'])
        if cfg['use_speaker']:
            smu, ssig, sco = fork(p, 'speaker_to_output', emb_speaker, ['gmm_mu', 'gmm_sigma', 'gmm_coeff'])
            mu, sigma, coeff = mu + smu, sigma + ssig, coeff + sco
        sigma = torch.exp(sigma) + cfg['epsilon']
        coeff = torch.softmax(coeff, -1) + cfg['epsilon']
        cost = cost_gmm(target, mu, sigma, coeff)
        next_x = mu
        extras.update(mu=mu, sigma=sigma)
    cost = (cost * mask).sum() / (mask.sum() + 1e-5)  # model.py:784

    new_carry = dict(h=list(h), k=k_all[-1], w=w_all[-1])  # model.py:786-791
    attention_vars = [next_x, k_all, w_all, coeff, phi_all, pi_all]
    extras.update(h=hs_all, ctx=ctx, readouts=readouts)
    return cost, new_carry, attention_vars, extras


def cost_and_grads_checkpointed(p, cfg, features, features_mask, labels, labels_mask, speaker=None, chunk=100,
                                pinned=None):
    """compute_cost(...) followed by cost.backward() for ONE window, with the memory of `chunk` steps: truncated
    nothing, recomputed everything (backpropagation through time with checkpoints at the chunk boundaries).  Pass 1
    walks the window chunk by chunk without a graph and keeps the carried state (h, kappa, w) entering each chunk;
    pass 2 walks the chunks backwards, rebuilds each chunk's graph from its stored carry, and backpropagates the
    chunk's share of the masked mean (model.py:784) plus <carried state out, gradient wrt it from the later chunks>.
    Parameter gradients accumulate in p[*].grad exactly as in the one-piece call (tests/test_oracle_cpu.py checks
    that at 1e-12); this exists so that the fp64 oracle fits the host memory at T_dec = 800 (model.py:726-737
    saves every step).  Returns (cost, attention_vars) with the per-chunk outputs concatenated along time.

    pinned (test infrastructure for long windows in a reduced-precision operand mode): a function `t -> carry` that hands
    over the state ENTERING step t as some other evaluation of the same window saved it (dict(h=[...], k=, w=) like
    `new_carry`).  Every chunk is then rebuilt from that state instead of from this function's own pass 1, so the
    backward pass differentiates the trajectory the other evaluation walked (re-pinned every `chunk` steps) while the
    chain of adjoints across the chunk boundaries stays this function's own.  That separates "is the 800-deep backward
    right" from "did two evaluations of an expansive forward map drift apart"."""
    T = features.shape[0] - 1
    bounds = list(range(0, T, chunk)) + [T]
    spans = list(zip(bounds[:-1], bounds[1:]))
    den = features_mask[1:].sum() + 1e-5
    carries, nums, avs = [None], [], []
    with torch.no_grad():
        carry = None
        for i, (a, b) in enumerate(spans):
            if pinned is not None and i > 0:
                carry = pinned(a)
                carries[-1] = carry
            c, carry, av, _ = compute_cost(p, cfg, features[a:b + 1], features_mask[a:b + 1], labels, labels_mask,
                                           speaker, 1 if i == 0 else 0, carry)
            carries.append(carry)
            nums.append(c * (features_mask[a + 1:b + 1].sum() + 1e-5))  # the chunk's masked sum
            avs.append(av)
    cost = sum(nums) / den
    dcarry = None
    for i in reversed(range(len(spans))):
        a, b = spans[i]
        cin = None
        if i > 0:
            src = carries[i]
            fresh = lambda x: x.detach().clone().requires_grad_()
            # (LSTM layers carry (state, cells) pairs, sampleRNN/lib/ops.py:505-553)
            cin = dict(h=[tuple(fresh(y) for y in x) if isinstance(x, tuple) else fresh(x) for x in src['h']],
                       k=fresh(src['k']), w=fresh(src['w']))
        c, cout, _, _ = compute_cost(p, cfg, features[a:b + 1], features_mask[a:b + 1], labels, labels_mask, speaker,
                                     1 if i == 0 else 0, cin)
        total = c * (features_mask[a + 1:b + 1].sum() + 1e-5) / den
        if dcarry is not None:
            for x, g in zip(cout['h'], dcarry['h']):
                for x_, g_ in (zip(x, g) if isinstance(x, tuple) else ((x, g),)):
                    if g_ is not None:
                        total = total + (x_ * g_).sum()
            total = total + (cout['k'] * dcarry['k']).sum() + (cout['w'] * dcarry['w']).sum()
        total.backward()
        if cin is not None:
            dcarry = dict(h=[tuple(y.grad for y in x) if isinstance(x, tuple) else x.grad for x in cin['h']],
                          k=cin['k'].grad, w=cin['w'].grad)
    cat = [torch.cat([av[j] for av in avs], 0) for j in range(len(avs[0]))]
    return cost, cat


def sample_gmm(mu, sigma, weight, unif, noise):
    """sample_gmm (model.py:94-118) with the randomness made explicit: Theano's
    theano_rng.multinomial(pvals=weight) one-hot draw + argmax == first k with cumsum(weight) > u
    (rng_mrg multinomial), epsilon = theano_rng.normal(...) == `noise`."""
    k = weight.shape[-1]
    dim = mu.shape[-1] // k  # model.py:97 (py2 integer division)
    mu = mu.reshape(-1, dim, k)
    sigma = sigma.reshape(-1, dim, k)
    cum = torch.cumsum(weight.reshape(-1, k), -1)
    idx = (cum > unif.reshape(-1, 1)).to(torch.int64).argmax(-1)
    idx = torch.where((cum > unif.reshape(-1, 1)).any(-1), idx, torch.full_like(idx, k - 1))
    ar = torch.arange(mu.shape[0])
    return mu[ar, :, idx] + sigma[ar, :, idx] * noise.reshape(-1, dim)


def sample_model(p, cfg, labels, labels_mask, speaker, num_steps, unif=None, noise=None):
    """Parrot.sample_model / sample_model_fun (model.py:826-1083).  MSE head:
    x_t = readout_to_output(readouts_t) is fed back deterministically (model.py:1010-1016).  GMM head
    (model.py:1017-1033): needs the explicit randomness unif [S,N], noise [S,N,O] (see sample_gmm).
    Returns [sample_x [S,N,O], k, w, pi, phi, pi_att] like the reference."""
    gmm = cfg['which_cost'] == 'GMM'
    assert not gmm or (unif is not None and noise is not None), 'GMM sampling needs explicit randomness'
    L, H, ln = cfg['num_layers'], cfg['rnn_h_dim'], cfg['layer_norm']
    N = labels.shape[0]
    dt = p['/parrot.initial_w'].dtype
    const = [[torch.zeros(N, wd, dtype=dt) for _, wd in layer_outs(cfg, l)] for l in range(1, L + 1)]
    spk_readout = spk_output = None
    if cfg['use_speaker']:  # model.py:846-874
        emb = p['/parrot/lookuptable.W'][speaker[:, 0]]
        spk_readout = linear(p, 'speaker_to_readout', emb)
        spk_output = (fork(p, 'speaker_to_output', emb, ['gmm_mu', 'gmm_sigma', 'gmm_coeff']) if gmm
                      else linear(p, 'speaker_to_output', emb))
        for l in range(1, L + 1):
            outs = fork(p, f'speaker_to_h{l}', emb[None], _names(cfg, l))
            const[l - 1] = [c_ + apply_norm(o_, ln)[0] for c_, o_ in zip(const[l - 1], outs)]
    ctx = encoder_apply(p, cfg, labels) * labels_mask[..., None]  # model.py:876-877
    init = initial_carry(p, cfg, N)  # always the learned initial states (model.py:1049-1054)
    h, w, k = init['h'], init['w'], init['k']
    x = torch.zeros(N, cfg['output_dim'], dtype=dt)  # model.py:834-835
    xs, ks, ws, phis, pis, cos = [], [], [], [], [], []
    for step in range(num_steps):
        seq_in = [[c_.clone() for c_ in c] for c in const]
        if cfg['weak_feedback']:  # model.py:899-908
            outs = fork(p, 'out_to_h1', x, _names(cfg, 1))
            seq_in[0] = [s_ + apply_norm(o_, ln) for s_, o_ in zip(seq_in[0], outs)]
        if cfg['full_feedback']:  # model.py:910-924
            for l in range(2, L + 1):
                outs = fork(p, f'out_to_h{l}', x, _names(cfg, l))
                seq_in[l - 1] = [s_ + apply_norm(o_, ln) for s_, o_ in zip(seq_in[l - 1], outs)]
        h, hout, k, w, phi, a = decoder_step(p, cfg, seq_in, h, k, w, ctx, sampling=True)
        readout = 0  # model.py:992-1006
        for l in range(1, L + 1):
            readout = readout + apply_norm(linear(p, f'h{l}_to_readout', hout[l - 1]), ln)
        readout = readout + linear(p, 'att_to_readout', w)
        if cfg['use_speaker']:
            readout = readout + spk_readout
        if not gmm:
            x = linear(p, 'readout_to_output', readout)  # model.py:1008-1013
            if cfg['use_speaker']:
                x = x + spk_output
            cos.append(x)
        else:  # model.py:1017-1033
            mu, sig, co = fork(p, 'readout_to_output', readout, ['gmm_mu', 'gmm_sigma', 'gmm_coeff'])
            if cfg['use_speaker']:
                mu, sig, co = mu + spk_output[0], sig + spk_output[1], co + spk_output[2]
            sig = torch.exp(sig - cfg['sampling_bias']) + cfg['epsilon']
            co = torch.softmax(co * (1. + cfg['sampling_bias']), -1) + cfg['epsilon']
            x = sample_gmm(mu, sig, co, unif[step].to(dt), noise[step].to(dt))
            cos.append(co)
        xs.append(x); ks.append(k); ws.append(w); phis.append(phi); pis.append(a)
    sx, kk, ww, ph, pa, cc = (torch.stack(v, 0) for v in (xs, ks, ws, phis, pis, cos))
    return [sx, kk, ww, cc, ph, pa]


# ----------------------------------------------------------------------------- optimiser
def clip_adam_step(params, grads, m, v, step, lr=1e-4, clip=9.0, b1=0.9, b2=0.999, eps=1e-8):
    """train.py:100-108: StepClipping(10*grad_clip) then Adam (Blocks defaults).
    All arguments are dicts name -> tensor; updates in place."""
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    scale = clip / tot if tot > clip else 1.0
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    for n in params:
        g = grads[n] * scale
        m[n].mul_(b1).add_(g, alpha=1 - b1)
        v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
        params[n].sub_(lr_t * m[n] / (v[n].sqrt() + eps))
    return tot
