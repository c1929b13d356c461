"""REFERENCE-EXECUTED golden vectors for the decoder and SampleRNN paths  ->  tests/golden/ref_golden.npz

Runs the reference's OWN sources -- /root/reference/model.py, sampleRNN/lib/ops.py, sampleRNN/lib/__init__.py and
sampleRNN/models/conditional/three_tier.py, read from the read-only checkout and executed unmodified
(oracle/refshim/loader.py) -- on eager torch-backed stand-ins for their two un-installable dependencies (Theano:
oracle/refshim/theano_shim.py; Blocks bricks: oracle/refshim/blocks_shim.py), in float64, on seeded inputs, and commits
the outputs.  This is what pins the oracle restatements (oracle/parrot_ref.py, oracle/samplernn_ref.py) and, through
them and directly, the HIP path:

  ops|*     lib.ops.Linear (weight norm, multi-input), __GRUStep, __LSTMStep, Embedding, softmax_and_argmax
            (ops.py:32-128, 252-297, 329-393, 461-553) called directly
  sr:*|*    three_tier.compute_cost (three_tier.py:534-636): cost, ip_cost, new_h0, new_big_h0 and d(cost+ip_cost)/d(every
            parameter) (torch autograd through the reference's own graph code), GRU-1 / LSTM-2 / GRU-2, reset 0 and 1;
            plus the reference's generate_and_save_samples loop (three_tier.py:750-851) with the temperature-0 sampler
  par:*|*   Parrot.compute_cost (model.py:551-824): cost, next_x, k, w, phi, pi_att, carry updates, every parameter
            gradient; a second TBPTT window with start_flag = 0; Parrot.sample_model_fun (model.py:826-1059)

What is NOT the reference here: the Theano op semantics and the Blocks brick algebra (Linear, Fork, LookupTable,
GatedRecurrent, Bidirectional) are restated from their published behaviour in the shims.  The GatedRecurrent restatement
is cross-checked against the reference's own twin __GRUStep (tests/test_oracle_cpu.py).

Regenerate (needs /root/reference):   python tests/golden/make_ref_golden.py
The case tables and input builders below are imported by the tests, which rebuild the same seeded inputs.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import make_batch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'ref_golden.npz')

# ----------------------------------------------------------------------------- SampleRNN cases
SR_DIM, SR_EMB = 32, 8
SR_CASES = {'gru1': ('GRU', 1), 'lstm2': ('LSTM', 2), 'gru2': ('GRU', 2)}
# stacks with Graves' skip connections (ops.py:650-695, 861-880; --skip_conn True, off in the reference's run string)
SR_SKIP_CASES = {'gru2skip': ('GRU', 2), 'lstm3skip': ('LSTM', 3)}
BIG_GRAD = 4096  # gradients with more elements are stored as 4 seeded random projections


def sr_inputs(rnn, n):
    """Seeded inputs of three_tier.compute_cost for a case (B=2, 160 target samples)."""
    g = torch.Generator().manual_seed(3)
    B, S_len, hm, D = 2, 160, (2 if rnn == 'LSTM' else 1), SR_DIM
    seq = torch.randint(0, 256, (B, S_len + 80), generator=g)
    feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
    mask = torch.ones(B, S_len + 80, dtype=torch.float64)
    mask[1, 200:] = 0
    h0 = torch.randn(B, n, hm * D, generator=g, dtype=torch.float64) * 0.3
    bh0 = torch.randn(B, n, hm * D, generator=g, dtype=torch.float64) * 0.3
    return seq, feats, h0, bh0, mask


def sr_params(rnn, n, skip=False):
    from oracle import samplernn_ref as S
    c = S.config(DIM=SR_DIM, EMB_SIZE=SR_EMB, RNN_TYPE=rnn, N_RNN=n, SKIP_CONN=skip)
    return c, S.init_params(c, seed=9, perturb=0.2)


def gen_features(T=4, B=3):
    g = torch.Generator().manual_seed(2)
    return torch.randn(T, B, 63, generator=g, dtype=torch.float64)


def project(name, g):
    """Compact stand-in for a big gradient: its products with 4 seeded random vectors (float64)."""
    g = torch.as_tensor(g).detach().double().reshape(-1)
    seed = int.from_bytes(name.encode()[-4:], 'little') % (2 ** 31)
    r = torch.randn(4, g.numel(), generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    return (r @ g).numpy()


def pack_grad(name, g):
    g = torch.as_tensor(g).detach().double()
    return g.numpy() if g.numel() <= BIG_GRAD else project(name, g)


# ----------------------------------------------------------------------------- Parrot cases (reference depth: 3 GRU layers)
PAR_SMALL = dict(rnn_h_dim=32, readouts_dim=24, encoder_dim=8, input_dim=12, speaker_dim=6, num_speakers=4,
                 encoder_type='bidirectional')
PAR_CASES = {
    'base': dict(),
    'fb_spk': dict(full_feedback=True, use_speaker=True),
    'softmax_ln': dict(attention_type='softmax', weak_feedback=True, layer_norm=True),
    'gmm': dict(which_cost='GMM', k_gmm=3, weak_feedback=True),
    'sharp': dict(weak_feedback=True, sharpening_coeff=1.3, timing_coeff=0.8, attention_alignment=0.7),
}
PAR_T, PAR_B, PAR_U, PAR_S = 7, 4, 9, 6


def par_setup(kw):
    from oracle import parrot_ref as R
    full = dict(PAR_SMALL, **kw)
    cfg = R.default_config(num_layers=3, **full)
    p = R.init_params(cfg, seed=11, scale_by_fan_in=True)
    batch = make_batch(cfg, PAR_T, PAR_B, PAR_U, seed=21, ragged=True, speaker=cfg['use_speaker'])
    return full, cfg, p, batch


# ----------------------------------------------------------------------------- generation
def _gen_ops(blob, ops, lib, th):
    """Operator-level vectors: the reference functions called directly on seeded inputs."""
    from oracle.refshim.loader import quiet_call as q
    g = torch.Generator().manual_seed(17)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    V = lambda t: th.Var(t.clone())  # noqa: E731
    lib.clear_all_params()
    D = 12
    x1, x2, h = rn(5, 7), rn(5, 3), rn(5, D) * 0.5
    hc = rn(5, 2 * D) * 0.5
    idx = torch.randint(0, 9, (5, 4), generator=g)
    logits = rn(3, 4, 6)
    logits[0, 0, 2] = logits[0, 0, 4] = 9.0  # a tie: argmax takes the lowest index

    def build():
        lin = ops.Linear('T.Lin', [7, 3], D, [V(x1), V(x2)], initialization='he', weightnorm=True)
        gru = getattr(ops, '__GRUStep')('T.GRU', D, D, lin, V(h), weightnorm=True)
        lstm = getattr(ops, '__LSTMStep')('T.LSTM', D, D, lin, V(hc), weightnorm=True)
        emb = ops.Embedding('T.Emb', 9, 5, V(idx.int()))
        am = ops.softmax_and_argmax(V(logits))
        return lin, gru, lstm, emb, am
    q(build)  # registers the parameters with the reference's own initialisers
    pg = torch.Generator().manual_seed(23)
    for name, v in sorted(lib._params.items()):  # seeded values (so that g != ||W|| and biases != 0)
        val = torch.randn(v.data.shape, generator=pg, dtype=torch.float64) * 0.4
        if name.split('.')[-1].startswith('g'):
            val = val.abs() + 0.5
        v.set_value(val.numpy())
        blob[f'ops|param:{name}'] = val.numpy()
    lin, gru, lstm, emb, am = q(build)
    for k, v in (('x1', x1), ('x2', x2), ('h', h), ('hc', hc), ('idx', idx), ('logits', logits)):
        blob[f'ops|in:{k}'] = v.numpy()
    for k, v in (('linear', lin), ('gru', gru), ('lstm', lstm), ('embedding', emb), ('argmax', am)):
        blob[f'ops|out:{k}'] = v.data.detach().numpy()
    lib.clear_all_params()


def _gen_sr(blob, lib, ops, tt, th):
    from oracle.refshim.loader import quiet_call as q
    V = lambda t: th.Var(t.clone())  # noqa: E731
    for case, (rnn, n) in list(SR_CASES.items()) + list(SR_SKIP_CASES.items()):
        skip = case in SR_SKIP_CASES
        lib.clear_all_params()
        tt.DIM = tt.BIG_DIM = SR_DIM
        tt.EMB_SIZE, tt.RNN_TYPE, tt.N_RNN, tt.N_BIG_RNN = SR_EMB, rnn, n, n
        tt.H0_MULT = 2 if rnn == 'LSTM' else 1
        tt.SKIP_CONN = skip
        c, p = sr_params(rnn, n, skip)
        seq, feats, h0, bh0, mask = sr_inputs(rnn, n)
        for reset in (0, 1):
            args = lambda: (V(seq.int()), V(feats), V(h0), V(bh0), V(torch.tensor(float(reset))), V(mask))  # noqa: E731
            q(tt.compute_cost, *args())  # registers the parameters
            assert set(lib._params) == set(p), set(lib._params) ^ set(p)
            for k, v in p.items():
                lib._params[k].set_value(v.numpy())
                lib._params[k].data.grad = None
            cost, ip_cost, allp, ipp, otherp, nh0, nbh0 = q(tt.compute_cost, *args())
            (cost.data + ip_cost.data).backward()
            pre = f'sr:{case}:r{reset}|'
            blob[pre + 'cost'] = cost.data.detach().numpy()
            blob[pre + 'ip_cost'] = ip_cost.data.detach().numpy()
            blob[pre + 'new_h0'] = nh0.data.detach().numpy()
            blob[pre + 'new_big_h0'] = nbh0.data.detach().numpy()
            blob[pre + 'n_params'] = np.array([len(allp), len(ipp), len(otherp)])
            for k in p:
                gr = lib._params[k].data.grad
                if gr is not None:
                    blob[pre + 'grad:' + k] = pack_grad(k, gr)
        if case == 'gru1':
            # the reference's own sample loop (three_tier.py:750-851); the three "compiled functions" it is handed are
            # thin closures over the reference's graph builders (theano.function cannot exist in an eager shim) and the
            # reference's temperature-0 sampler (ops.py:282-286, 296-297) replaces its MRG multinomial draw
            def big_fn(s, bh, reset, f):
                o = q(tt.big_frame_level_rnn, V(torch.from_numpy(s)), V(torch.from_numpy(bh).double()),
                      V(torch.tensor(float(reset))), V(torch.from_numpy(f).double()))
                return o[0].data.detach().numpy(), o[1].data.detach().numpy()

            def frame_fn(s, big_out, h, reset):
                o = q(tt.frame_level_rnn, V(torch.from_numpy(s)), V(torch.from_numpy(big_out)).dimshuffle(0, 'x', 1),
                      V(torch.from_numpy(h).double()), V(torch.tensor(float(reset))))
                return o[0].data.detach().numpy(), o[1].data.detach().numpy()

            def sample_fn(fo, prev):
                lg = q(tt.sample_level_predictor, V(torch.from_numpy(fo)), V(torch.from_numpy(prev)))
                return q(ops.softmax_and_argmax, lg).data.numpy()
            import tempfile
            feats_g = gen_features()
            refdir = os.environ.get('PARROT_REFERENCE', '/root/reference')
            sys.path.insert(0, refdir)  # the loop does `from quantize import mu2linear` (three_tier.py:846)
            import quantize as ref_quantize
            seen = []
            real_mu = ref_quantize.mu2linear

            def recording_mu2linear(samp):  # the loop has no return value: record the integer samples it decodes
                seen.append(np.array(samp, dtype=np.int32))
                return real_mu(samp)
            ref_quantize.mu2linear = recording_mu2linear
            try:
                with tempfile.TemporaryDirectory() as tmp:
                    q(tt.generate_and_save_samples, 'g', path_to_save=tmp, features=feats_g.numpy(),
                      features_length=None, big_frame_level_generate_fn=big_fn, frame_level_generate_fn=frame_fn,
                      sample_level_generate_fn=sample_fn)
            finally:
                ref_quantize.mu2linear = real_mu
                sys.path.remove(refdir)
            blob['sr:gru1|samples'] = np.stack(seen)  # [3, 320] greedy indices, first 80 = Q_ZERO
    lib.clear_all_params()


def _gen_par(blob, M, th):
    from oracle import parrot_ref as R
    from oracle.refshim.loader import quiet_call as q
    V = lambda t: th.Var(t.clone())  # noqa: E731
    for case, kw in PAR_CASES.items():
        full, cfg, p, (feat, fm, lab, lm, spk) = par_setup(kw)
        ref = M.Parrot(name='parrot', **full)
        ref.allocate()
        named = ref.named_parameters()
        named['/parrot.initial_w'] = ref.initial_w
        assert set(named) == set(p), set(named) ^ set(p)
        for k, v in p.items():
            named[k].set_value(v.numpy())
        if cfg['which_cost'] == 'GMM':
            # next_x = sample_gmm(...) is built unconditionally (model.py:781); its draw does not enter the cost.
            M.sample_gmm = lambda mu, sigma, weight, rng: mu[..., :cfg['output_dim']]
        sv = None if spk is None else V(spk.int())
        pre = f'par:{case}|'
        # ---- window 1 (start_flag = 1) on frames [0, 5], window 2 (start_flag = 0) on frames [4, T]
        cost, updates, av, _ = q(ref.compute_cost, V(feat[:5]), V(fm[:5]), V(lab.int()), V(lm), sv,
                                 V(torch.tensor(1.0)), PAR_B)
        blob[pre + 'w1:cost'] = cost.data.detach().numpy()
        carry = [u[1].data.detach().clone() for u in updates]  # h1, h2, h3, k, w (model.py:786-791)
        orig = ref.initial_states

        def carried(batch):  # what theano.function's `updates` would have left in the last_* shared variables
            v = list(orig(batch))
            for slot, c in zip((1, 3, 5, 9, 7), carry):
                v[slot] = V(c)
            return tuple(v)
        ref.initial_states = carried
        cost2, _, av2, _ = q(ref.compute_cost, V(feat[4:]), V(fm[4:]), V(lab.int()), V(lm), sv, V(torch.tensor(0.0)),
                             PAR_B)
        ref.initial_states = orig
        blob[pre + 'w2:cost'] = cost2.data.detach().numpy()
        if cfg['which_cost'] == 'MSE':
            blob[pre + 'w2:next_x'] = av2[0].data.detach().numpy()
        blob[pre + 'w2:k'] = av2[1].data.detach().numpy()
        # ---- the full window with gradients
        cost, updates, av, _ = q(ref.compute_cost, V(feat), V(fm), V(lab.int()), V(lm), sv, V(torch.tensor(1.0)), PAR_B)
        cost.data.backward()
        blob[pre + 'cost'] = cost.data.detach().numpy()
        names = ('next_x', 'k', 'w', 'coeff', 'phi', 'pi_att')
        for i, n in enumerate(names):
            if cfg['which_cost'] == 'GMM' and n == 'next_x':
                continue
            blob[pre + n] = av[i].data.detach().numpy()
        for n, u in zip(('h1', 'h2', 'h3', 'k', 'w'), updates):
            blob[pre + 'carry:' + n] = u[1].data.detach().numpy()
        for k in p:
            gr = named[k].data.grad
            blob[pre + 'grad:' + k] = pack_grad(k, torch.zeros_like(named[k].data) if gr is None else gr)
        # ---- decode (MSE head only: the GMM head draws from Theano's MRG stream)
        if cfg['which_cost'] == 'MSE':
            out = q(ref.sample_model_fun, V(lab.int()), V(lm), sv, PAR_B, PAR_S)
            for n, v in zip(('sample_x', 'k', 'w', 'pi', 'phi', 'pi_att'), out[:6]):
                blob[pre + 'sample:' + n] = v.data.detach().numpy()


def generate():
    from oracle.refshim import loader, theano_shim as th
    loader.install('float64')
    lib, ops, tt = loader.load_sample_rnn()
    M = loader.load_model()
    blob = {}
    with torch.enable_grad():
        _gen_ops(blob, ops, lib, th)
        _gen_sr(blob, lib, ops, tt, th)
        _gen_par(blob, M, th)
    return blob


if __name__ == '__main__':
    b = generate()
    np.savez_compressed(PATH, **b)
    print(f"{len(b)} arrays -> {PATH} ({os.path.getsize(PATH) / 1024:.0f} KiB)")
