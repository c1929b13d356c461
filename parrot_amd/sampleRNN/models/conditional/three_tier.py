"""Conditional three-tier SampleRNN -- mirrors reference sampleRNN/models/conditional/three_tier.py:
the tier builders (``big_frame_level_rnn`` :291-380, ``frame_level_rnn`` :382-450,
``sample_level_predictor`` :452-515), ``compute_cost`` (:534-636), ``getting_generation_functions``
(:703-734) and ``generate_and_save_samples`` (:750-851), with the run configuration the reference
hard-codes at import time (:145-202) as module globals.

Differences by construction (no Theano graph): the functions compute eagerly on GPU tensors through
the HIP library; ``configure(...)`` can override the hard-coded hyper-parameters (tests use small
dimensions); generation has a device-resident fast path (``DeviceGenerator`` ->
samplernn_generate_*, include/parrot_hip.h) that ``generate_and_save_samples`` uses.
"""
from __future__ import annotations

import ctypes as C
import os
from time import time

import numpy
import torch

from .... import _lib
from .... import ops as hip
from ... import lib
from ...lib import ops as lops

# --- three_tier.py:145-202: the reference calls get_args() on a hard-coded string at import time
SEQ_LEN = 4000
BIG_FRAME_SIZE = 80
FRAME_SIZE = 10
OVERLAP = BIG_FRAME_SIZE
WEIGHT_NORM = True
EMB_SIZE = 256
SKIP_CONN = False
DIM = 1024
BIG_DIM = DIM
N_RNN = 1
N_BIG_RNN = N_RNN
RNN_TYPE = 'GRU'
H0_MULT = 2 if RNN_TYPE == 'LSTM' else 1
LEARN_H0 = True
Q_LEVELS = 256
Q_TYPE = 'mu-law'
BATCH_SIZE = 8
GRAD_CLIP = 1
BITRATE = 16000
TEMPERATURE = 1.
FEAT_DIM = 63
Q_ZERO = numpy.int32(Q_LEVELS // 2)


def configure(**kw):
    """Overrides the hard-coded run configuration (extension; the reference edits the source)."""
    g = globals()
    for k, v in kw.items():
        assert k in g, k
        g[k] = v
    g['BIG_DIM'] = g['DIM']
    g['N_BIG_RNN'] = g['N_RNN']
    g['H0_MULT'] = 2 if g['RNN_TYPE'] == 'LSTM' else 1
    g['Q_ZERO'] = numpy.int32(g['Q_LEVELS'] // 2)
    g['OVERLAP'] = g['BIG_FRAME_SIZE']


def _frames_to_float(frames):
    """three_tier.py:309-310, 398-399."""
    f = (frames.to(torch.float32) / lib.floatX(Q_LEVELS // 2)) - lib.floatX(1)
    return f * lib.floatX(2)


def _stacked(name, n_rnn, in_dim, dim, inp, h0):
    if RNN_TYPE == 'GRU':
        return lops.stackedGRU(name + '.GRU', n_rnn, in_dim, dim, inp, h0, WEIGHT_NORM, SKIP_CONN)
    return lops.stackedLSTM(name + '.LSTM', n_rnn, in_dim, dim, inp, h0, WEIGHT_NORM, SKIP_CONN)


def big_frame_level_rnn(input_sequences, h0, reset, features):
    """three_tier.py:291-380: (output [B, 8n, DIM], last_hidden, independent_preds [B, 80n, Q])."""
    B = input_sequences.shape[0]
    frames = _frames_to_float(input_sequences.reshape(B, -1, BIG_FRAME_SIZE))
    rnn_inp = lops.Linear('BigFrameLevel.rnn_inp_fusion', [BIG_FRAME_SIZE, FEAT_DIM], BIG_DIM,
                          [frames, features.to(torch.float32)], initialization='he', weightnorm=WEIGHT_NORM)
    learned_h0 = lib.param('BigFrameLevel.h0', numpy.zeros((N_BIG_RNN, H0_MULT * BIG_DIM), dtype='float32'),
                           trainable=LEARN_H0)
    if reset:
        h0 = learned_h0.unsqueeze(0).expand(B, -1, -1)
    rnns_out, last_hidden = _stacked('BigFrameLevel', N_BIG_RNN, BIG_DIM, BIG_DIM, rnn_inp, h0)
    output = lops.Linear('BigFrameLevel.Output', BIG_DIM, DIM * BIG_FRAME_SIZE // FRAME_SIZE, rnns_out,
                         initialization='he', weightnorm=WEIGHT_NORM)
    output = output.reshape(B, output.shape[1] * BIG_FRAME_SIZE // FRAME_SIZE, DIM)
    independent_preds = lops.Linear('BigFrameLevel.IndependentPreds', BIG_DIM, Q_LEVELS * BIG_FRAME_SIZE, rnns_out,
                                    initialization='he', weightnorm=WEIGHT_NORM)
    independent_preds = independent_preds.reshape(B, independent_preds.shape[1] * BIG_FRAME_SIZE, Q_LEVELS)
    return output, last_hidden, independent_preds


def frame_level_rnn(input_sequences, other_input, h0, reset):
    """three_tier.py:382-450: (output [B, 10m, DIM], last_hidden)."""
    B = input_sequences.shape[0]
    frames = _frames_to_float(input_sequences.reshape(B, -1, FRAME_SIZE))
    gru_input = lops.Linear('FrameLevel.InputExpand', FRAME_SIZE, DIM, frames, initialization='he',
                            weightnorm=WEIGHT_NORM) + other_input
    learned_h0 = lib.param('FrameLevel.h0', numpy.zeros((N_RNN, H0_MULT * DIM), dtype='float32'), trainable=LEARN_H0)
    if reset:
        h0 = learned_h0.unsqueeze(0).expand(B, -1, -1)
    rnns_out, last_hidden = _stacked('FrameLevel', N_RNN, DIM, DIM, gru_input, h0)
    output = lops.Linear('FrameLevel.Output', DIM, FRAME_SIZE * DIM, rnns_out, initialization='he',
                         weightnorm=WEIGHT_NORM)
    output = output.reshape(B, output.shape[1] * FRAME_SIZE, DIM)
    return output, last_hidden


def sample_level_predictor(frame_level_outputs, prev_samples):
    """three_tier.py:452-515: logits [rows, Q_LEVELS]."""
    assert EMB_SIZE > 0, 'no support for one-hot in three_tier (three_tier.py:458)'
    # Same parameters, created in the reference's order (Embedding, L1_PrevSamples, L2, L3, Output); the arithmetic runs
    # on the fused HIP operators (round 5):
    #   * Embedding -> reshape -> L1_PrevSamples (no bias) -> + frame_level_outputs is, row by row, a sum of FRAME_SIZE rows
    #     of the folded table (Embedding . W1_j) plus the frame-tier row: hip.embed_sum (gather-sum forward, segmented-sum
    #     backward; no [rows, FRAME_SIZE * EMB_SIZE] activation and no K = FRAME_SIZE * EMB_SIZE product);
    #   * relu(L2) -> relu(L3) -> Output: hip.relu_mlp (bias + ReLU in the products' epilogues, the ReLU masks of the
    #     backward pass in the epilogues of the dx products).
    vectors = lops.Embedding('SampleLevel.Embedding', Q_LEVELS, EMB_SIZE, None)
    lops.Linear('SampleLevel.L1_PrevSamples', FRAME_SIZE * EMB_SIZE, DIM, None, biases=False, initialization='he',
                weightnorm=WEIGHT_NORM, just_params=True)
    out = hip.embed_sum(vectors, lops.effective_weight('SampleLevel.L1_PrevSamples', 0, WEIGHT_NORM),
                        prev_samples.reshape(-1, FRAME_SIZE), frame_level_outputs.reshape(-1, DIM))
    wb = []
    for name, dout, init in (('SampleLevel.L2', DIM, 'he'), ('SampleLevel.L3', DIM, 'he'), ('SampleLevel.Output', Q_LEVELS, None)):
        lops.Linear(name, DIM, dout, None, initialization=init, weightnorm=WEIGHT_NORM, just_params=True)
        wb += [lops.effective_weight(name, 0, WEIGHT_NORM), lib.param(name + '.b')]
    return hip.relu_mlp(out, *wb)


def compute_cost(sequences, features, h0, big_h0, reset, mask):
    """three_tier.py:534-636.  sequences [B, S+80] int, features [B, S/80, 63], mask [B, S+80].
    Returns (cost, ip_cost, all_params, ip_params, other_params, new_h0, new_big_h0); costs in bits."""
    big_input_sequences = sequences[:, :-BIG_FRAME_SIZE]
    input_sequences = sequences[:, BIG_FRAME_SIZE - FRAME_SIZE:-FRAME_SIZE]
    target_sequences = sequences[:, BIG_FRAME_SIZE:]
    target_mask = mask[:, BIG_FRAME_SIZE:].to(torch.float32)
    big_frame_level_outputs, new_big_h0, big_frame_independent_preds = \
        big_frame_level_rnn(big_input_sequences, big_h0, reset, features)
    frame_level_outputs, new_h0 = frame_level_rnn(input_sequences, big_frame_level_outputs, h0, reset)
    prev_samples = sequences[:, BIG_FRAME_SIZE - FRAME_SIZE:-1]
    prev_samples = prev_samples.unfold(1, FRAME_SIZE, 1).reshape(-1, FRAME_SIZE)  # images2neibs, :555-558
    sample_level_outputs = sample_level_predictor(frame_level_outputs.reshape(-1, DIM), prev_samples)
    tgt = target_sequences.reshape(-1, 1).long()
    log2e = float(numpy.log2(numpy.e))

    def ce_bits(logits):
        ce = hip.softmax_ce(logits, tgt)  # logsumexp - picked logit, one HIP pass (and one for its gradient)
        ce = ce.reshape(target_sequences.shape) * target_mask
        return ce.sum() / (target_mask.sum() + 1e-5) * log2e

    cost = ce_bits(sample_level_outputs)
    ip_cost = ce_bits(big_frame_independent_preds.reshape(-1, Q_LEVELS))
    all_named = lib.named_params()
    # three_tier.py:595-600: ip_params = the BigFrameLevel parameters the independent-prediction cost is a function of
    # (everything in that tier except its Output projection, which only feeds the frame tier); other_params = the
    # remaining parameters of `cost`; all_params = ip_params + other_params (so IndependentPreds is in, via ip_params).
    trainable = [(n, p) for n, p in all_named.items() if getattr(p, 'param', False)]
    ip_params = [p for n, p in trainable if 'BigFrameLevel' in n and not n.startswith('BigFrameLevel.Output.')]
    other_params = [p for n, p in trainable if 'BigFrameLevel' not in n or n.startswith('BigFrameLevel.Output.')]
    all_params = ip_params + other_params
    return cost, ip_cost, all_params, ip_params, other_params, new_h0, new_big_h0


# ----------------------------------------------------------------------------- generation
def getting_generation_functions(sequences=None, h0=None, big_h0=None, reset=None, features=None):
    """three_tier.py:703-734.  Returns (big_frame_level_generate_fn, frame_level_generate_fn,
    sample_level_generate_fn) taking / returning numpy arrays like the compiled Theano functions."""
    dev = lib.device()

    def _t(x, dtype=None):
        t = torch.as_tensor(numpy.asarray(x)).to(dev)
        return t.to(dtype) if dtype is not None else t

    def big_fn(seq, big_h0_, reset_, feats):
        with torch.no_grad():
            out, last, _ = big_frame_level_rnn(_t(seq), _t(big_h0_, torch.float32), bool(reset_), _t(feats, torch.float32))
        return out.cpu().numpy(), last.cpu().numpy()

    def frame_fn(seq, big_out, h0_, reset_):
        with torch.no_grad():
            out, last = frame_level_rnn(_t(seq), _t(big_out, torch.float32).unsqueeze(1), _t(h0_, torch.float32),
                                        bool(reset_))
        return out.cpu().numpy(), last.cpu().numpy()

    def sample_fn(frame_out, prev, temperature=1.0):
        with torch.no_grad():
            logits = sample_level_predictor(_t(frame_out, torch.float32), _t(prev))
            return lops.softmax_and_sample(logits, temperature=temperature).cpu().numpy().astype('int32')

    return big_fn, frame_fn, sample_fn


class DeviceGenerator:
    """Device-resident generation loop (samplernn_generate_*, include/parrot_hip.h)."""

    def __init__(self, batch, n_frames, temperature=0.0, seed=234, use_graph=True):
        """GRU or LSTM tiers, 1..5 stacked layers (three_tier.py:147-169).  Stacks with skip connections run on the literal
        three-function loop (generate_and_save_samples(..., use_device_loop=False) is chosen automatically)."""
        if SKIP_CONN:
            raise NotImplementedError("the device-resident generator does not take skip-connection stacks")
        self.B, self.T = batch, n_frames
        dev = lib.device()
        f = dict(device=dev, dtype=torch.float32)
        D, FS, BFS, Q = DIM, FRAME_SIZE, BIG_FRAME_SIZE, Q_LEVELS
        nfr = BFS // FS
        ew = lambda n, i=0: lops.effective_weight(n, i, WEIGHT_NORM).detach().contiguous()
        pb = lambda n: lib.param(n + '.b').detach().contiguous()
        with torch.no_grad():
            single = RNN_TYPE == 'GRU' and N_RNN == 1
            w = dict(
                big_Win_frames=ew('BigFrameLevel.rnn_inp_fusion', 0), big_Win_feats=ew('BigFrameLevel.rnn_inp_fusion', 1),
                big_bin=pb('BigFrameLevel.rnn_inp_fusion'),
                big_Wout=ew('BigFrameLevel.Output'), big_bout=pb('BigFrameLevel.Output'),
                frm_Win=ew('FrameLevel.InputExpand'), frm_bin=pb('FrameLevel.InputExpand'),
                frm_Wout=ew('FrameLevel.Output'), frm_bout=pb('FrameLevel.Output'),
                W2=ew('SampleLevel.L2'), b2=pb('SampleLevel.L2'), W3=ew('SampleLevel.L3'), b3=pb('SampleLevel.L3'),
                W4=ew('SampleLevel.Output'), b4=pb('SampleLevel.Output'))
            # fold Embedding . L1_PrevSamples into a [FS, Q, D] table (one GEMM per previous-sample position)
            emb = lib.param('SampleLevel.Embedding').detach().contiguous()
            W1 = ew('SampleLevel.L1_PrevSamples')
            tbl = torch.empty(FS, Q, D, **f)
            for pos in range(FS):
                hip.gemm(emb, W1[pos * EMB_SIZE:(pos + 1) * EMB_SIZE], out=tbl[pos])
            w['emb_tbl'] = tbl
            # the RNN stacks of the two tiers: per layer (Input.W, Input.b | b, Recurrent_Gates, Recurrent_Candidate | -)
            self.layers = {}
            for tier, tag in (('BigFrameLevel', 'big'), ('FrameLevel', 'frm')):
                for k in range(1, N_RNN + 1):
                    if RNN_TYPE == 'GRU':
                        pre = f'{tier}.GRU{k}.Step'
                        self.layers[(tag, k - 1)] = [ew(pre + '.Input'), pb(pre + '.Input'), ew(pre + '.Recurrent_Gates'),
                                                     ew(pre + '.Recurrent_Candidate')]
                    else:
                        pre = f'{tier}.LSTM{k}.Step'
                        self.layers[(tag, k - 1)] = [ew(pre + '.Input'), lib.param(pre + '.b').detach().contiguous(),
                                                     ew(pre + '.Recurrent_Gates')]
            if single:
                for tag in ('big', 'frm'):
                    U_, bU_, Wg_, Wc_ = self.layers[(tag, 0)]
                    w.update({f'{tag}_U': U_, f'{tag}_bU': bU_, f'{tag}_Wg': Wg_, f'{tag}_Wc': Wc_})
        self.w = w
        self.single = single
        self.ws = dict(
            samples=torch.zeros(batch, BFS * n_frames, device=dev, dtype=torch.int32),
            features=torch.zeros(n_frames, batch, FEAT_DIM, **f),
            big_h=torch.zeros(batch, D, **f), frm_h=torch.zeros(batch, D, **f),
            xf_big=torch.zeros(batch, BFS, **f), xf_frm=torch.zeros(batch, FS, **f),
            feat_cur=torch.zeros(batch, FEAT_DIM, **f), gru_in=torch.zeros(batch, D, **f),
            P=torch.zeros(batch, 4 * D, **f), z=torch.zeros(batch, D, **f), r=torch.zeros(batch, D, **f),
            rh=torch.zeros(batch, D, **f), big_out=torch.zeros(batch, nfr * D, **f),
            frame_out=torch.zeros(batch, FS * D, **f), o1=torch.zeros(batch, D, **f), o2=torch.zeros(batch, D, **f),
            o3=torch.zeros(batch, D, **f), logits=torch.zeros(batch, Q, **f),
            tbase=torch.zeros(1, device=dev, dtype=torch.int32))
        d = _lib.SampleRnnGenDesc()
        d.B, d.D, d.T, d.Q, d.FS, d.BFS, d.feat_dim, d.use_graph = batch, D, n_frames, Q, FS, BFS, FEAT_DIM, int(use_graph)
        d.temperature, d.seed = float(temperature), int(seed)
        for k, v in list(w.items()) + list(self.ws.items()):
            setattr(d, k, v.data_ptr())
        if not self.single:
            lstm = RNN_TYPE == 'LSTM'
            d.n_rnn, d.lstm = N_RNN, int(lstm)
            self.states = {}
            for tag in ('big', 'frm'):
                for k in range(N_RNN):
                    for q, t in enumerate(self.layers[(tag, k)]):
                        getattr(d, tag + '_L')[k][q] = t.data_ptr()
                    self.states[(tag, 'h', k)] = torch.zeros(batch, D, **f)
                    getattr(d, tag + '_hs')[k] = self.states[(tag, 'h', k)].data_ptr()
                    if lstm:
                        self.states[(tag, 'c', k)] = torch.zeros(batch, D, **f)
                        getattr(d, tag + '_cs')[k] = self.states[(tag, 'c', k)].data_ptr()
            self.ws['gate_ws'] = torch.zeros(batch, 4 * D, **f)
            self.ws['layer_tmp'] = torch.zeros(batch, D, **f)
            d.gate_ws, d.layer_tmp = self.ws['gate_ws'].data_ptr(), self.ws['layer_tmp'].data_ptr()
        # persistent-thread sample kernel (sr_persist.hip): zero-filled exchange / barrier workspace when the shape qualifies
        n = int(_lib.load().samplernn_persist_floats(C.byref(d)))
        if n > 0:
            self.ws['persist_ws'] = torch.zeros(n, **f)
            d.persist_ws, d.persist_ws_floats = self.ws['persist_ws'].data_ptr(), n
        self.desc = d
        self.plan = C.c_void_p()
        _lib.call('samplernn_generate_create', C.byref(d), C.byref(self.plan))
        self.persistent = bool(_lib.load().samplernn_generate_is_persistent(self.plan))

    def generate(self, features):
        """features [T, B, 63] time-major (what generate_and_save_samples receives) -> samples [B, 80*T] int32."""
        ws = self.ws
        ws['features'].copy_(torch.as_tensor(features).to(ws['features'].device, torch.float32))
        ws['samples'].zero_()
        ws['samples'][:, :BIG_FRAME_SIZE] = int(Q_ZERO)  # three_tier.py:795
        # reset = (t == BIG_FRAME_SIZE): the first step of both tiers starts from the learned h0 (:334-342, 411-419)
        if self.single:
            ws['big_h'].copy_(lib.param('BigFrameLevel.h0').detach()[0].unsqueeze(0).expand(self.B, -1))
            ws['frm_h'].copy_(lib.param('FrameLevel.h0').detach()[0].unsqueeze(0).expand(self.B, -1))
        else:  # h0 [N_RNN, H0_MULT * D]: LSTM layers carry [s | c] (ops.py:552)
            for tier, tag in (('BigFrameLevel', 'big'), ('FrameLevel', 'frm')):
                h0 = lib.param(tier + '.h0').detach()
                for k in range(N_RNN):
                    self.states[(tag, 'h', k)].copy_(h0[k, :DIM].unsqueeze(0).expand(self.B, -1))
                    if RNN_TYPE == 'LSTM':
                        self.states[(tag, 'c', k)].copy_(h0[k, DIM:].unsqueeze(0).expand(self.B, -1))
        _lib.call('samplernn_generate_run', self.plan, hip._stream())
        if self.persistent:  # a team of the persistent kernel that gave up leaves invalid samples: fail loudly
            _lib.call('samplernn_generate_status', self.plan)
        return ws['samples']

    def close(self):
        if self.plan:
            _lib.load().samplernn_generate_destroy(self.plan)
            self.plan = None


def write_audio_file(name, data, path_to_save):
    """three_tier.py:739-748."""
    import scipy.io.wavfile
    data = data.astype('float32')
    data -= data.min()
    data /= data.max()
    data -= 0.5
    data *= 0.95
    scipy.io.wavfile.write(os.path.join(path_to_save, name + '.wav'), BITRATE, data)


def generate_and_save_samples(tag, path_to_save=None, features=None, features_length=None, noise_level=0.,
                              big_frame_level_generate_fn=None, frame_level_generate_fn=None,
                              sample_level_generate_fn=None, npy_address=None, temperature=TEMPERATURE,
                              use_device_loop=True):
    """three_tier.py:750-851.  With use_device_loop (default) the per-sample loop runs on the GPU
    (DeviceGenerator); otherwise the reference's three-function Python loop is executed literally."""
    total_time = time()
    if npy_address is not None:
        test_feats = numpy.load(npy_address).astype('float32')  # [T,B,63]
    else:
        assert features is not None
        test_feats = numpy.asarray(features, dtype='float32')
    n_seqs, n_frames = test_feats.shape[1], test_feats.shape[0]
    LENGTH = n_frames * BIG_FRAME_SIZE
    if use_device_loop and not SKIP_CONN:  # (the device loop does not take the skip-connection stacks: literal loop)
        gen = DeviceGenerator(n_seqs, n_frames, temperature=temperature)
        samples = gen.generate(test_feats).cpu().numpy()
        gen.close()
    else:
        feats_bt = test_feats.transpose(1, 0, 2)
        samples = numpy.zeros((n_seqs, LENGTH), dtype='int32')
        samples[:, :BIG_FRAME_SIZE] = Q_ZERO
        big_h0 = numpy.zeros((n_seqs, N_BIG_RNN, H0_MULT * BIG_DIM), dtype='float32')
        h0 = numpy.zeros((n_seqs, N_RNN, H0_MULT * DIM), dtype='float32')
        big_out = frame_out = None
        for t in range(BIG_FRAME_SIZE, LENGTH):
            if t % BIG_FRAME_SIZE == 0:
                big_out, big_h0 = big_frame_level_generate_fn(
                    samples[:, t - BIG_FRAME_SIZE:t], big_h0, numpy.int32(t == BIG_FRAME_SIZE),
                    feats_bt[:, t // BIG_FRAME_SIZE, :][:, None, :])
            if t % FRAME_SIZE == 0:
                frame_out, h0 = frame_level_generate_fn(
                    samples[:, t - FRAME_SIZE:t], big_out[:, (t // FRAME_SIZE) % (BIG_FRAME_SIZE // FRAME_SIZE)],
                    h0, numpy.int32(t == BIG_FRAME_SIZE))
            samples[:, t] = sample_level_generate_fn(frame_out[:, t % FRAME_SIZE], samples[:, t - FRAME_SIZE:t],
                                                     temperature)
    total_time = time() - total_time
    print("{} samples of {} seconds length generated in {} seconds.".format(n_seqs, LENGTH / float(BITRATE), total_time))
    if path_to_save is not None:
        os.makedirs(path_to_save, exist_ok=True)
        for i in range(n_seqs):
            samp = samples[i, :int(features_length[i]) * 80] if features_length is not None else samples[i]
            if Q_TYPE == 'mu-law':
                samp = hip.mu2linear(torch.from_numpy(samp.astype('int32')).to(lib.device())).cpu().numpy()
            write_audio_file("sample_{}_{}".format(tag, i), samp, path_to_save)
    return samples
