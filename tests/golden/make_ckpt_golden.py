"""REFERENCE-WRITTEN SampleRNN checkpoint  ->  tests/golden/ref_samplernn_ckpt.pkl

Row f3 of the scope contract (checkpoint bridge).  The reference's own `lib.save_params` / `lib.load_params`
(/root/reference/sampleRNN/lib/__init__.py:96-109) are EXECUTED (oracle/refshim/loader.py) on the parameter registry its
own `three_tier.compute_cost` (three_tier.py:534-636) fills -- the reference's dotted names and shapes, tiny widths:

  1. the reference registers its parameters, they get seeded values, the reference's save_params writes the pickle
     committed here (Q_LEVELS lowered to 16 to keep the fixture small: names and ranks are unaffected);
  2. the other direction, checked when this script runs: parrot_amd.sampleRNN.lib.save_params writes a pickle, the
     reference's load_params reads it into its own registry, and every value must come back bit for bit.

tests/test_ckpt_cpu.py loads the committed pickle with the PRODUCT's load_params and, when /root/reference is present
(this container), repeats direction 2.  The Blocks `.tar` half of the bridge (train.py:157-173, sample.py:39-42, 83)
cannot be pinned this way: blocks.serialization is not in /root/reference, so parrot_amd/checkpoint.py stays validated
against its own writer and the documented layout only.

The pickle is written by Python 3's pickle (the reference runs `cPickle.dump(param_vals, f)` under Python 2: same
{name: ndarray} dict, protocol 0 text there); loaders must therefore accept latin1-encoded Python-2 pickles too, which
parrot_amd.sampleRNN.lib.load_params does (`encoding='latin1'`).

Regenerate (needs /root/reference):   python tests/golden/make_ckpt_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HERE = os.path.dirname(os.path.abspath(__file__))
PKL = os.path.join(HERE, 'ref_samplernn_ckpt.pkl')
DIM, EMB, RNN, N, Q = 16, 8, 'GRU', 2, 16


def reference_registry():
    """The reference's lib / three_tier with the registry filled by its own compute_cost; seeded float32 values."""
    from oracle.refshim import loader
    from oracle.refshim.loader import quiet_call as q
    lib, ops, tt = loader.load_sample_rnn()
    from oracle.refshim import theano_shim as th
    V = lambda t: th.Var(t.clone())  # noqa: E731
    lib.clear_all_params()
    tt.DIM = tt.BIG_DIM = DIM
    tt.EMB_SIZE, tt.RNN_TYPE, tt.N_RNN, tt.N_BIG_RNN, tt.H0_MULT, tt.SKIP_CONN = EMB, RNN, N, N, 1, False
    tt.Q_LEVELS, tt.Q_ZERO = Q, Q // 2
    g = torch.Generator().manual_seed(3)
    B, S_len = 2, 160
    seq = torch.randint(0, Q, (B, S_len + 80), generator=g)
    feats = torch.randn(B, S_len // 80, 63, generator=g, dtype=torch.float64)
    h0 = torch.zeros(B, N, DIM, dtype=torch.float64)
    mask = torch.ones(B, S_len + 80, dtype=torch.float64)
    q(tt.compute_cost, V(seq.int()), V(feats), V(h0), V(h0.clone()), V(torch.tensor(1.0)), V(mask))
    pg = torch.Generator().manual_seed(41)
    vals = {}
    for name, v in sorted(lib._params.items()):
        val = (torch.randn(tuple(v.get_value().shape), generator=pg) * 0.3).numpy().astype(np.float32)
        v.set_value(val)
        vals[name] = val
    return lib, vals


def product_roundtrip_through_reference(lib, vals):
    """Direction 2: product save_params -> the reference's load_params -> the reference's registry."""
    from parrot_amd.sampleRNN import lib as plib
    plib.delete_all_params()
    plib.set_device('cpu')
    rng = np.random.RandomState(7)
    fresh = {n: rng.randn(*v.shape).astype(np.float32) for n, v in vals.items()}
    for n, v in fresh.items():
        plib.param(n, v)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'product.pkl')
        plib.save_params(path)
        lib.load_params(path)  # the reference's own loader
    for n, v in fresh.items():
        got = np.asarray(lib._params[n].get_value())
        assert got.shape == v.shape and np.array_equal(got.astype(np.float32), v), n
    plib.delete_all_params()
    return len(fresh)


def main():
    lib, vals = reference_registry()
    lib.save_params(PKL)  # the reference's own writer
    n = product_roundtrip_through_reference(lib, vals)
    print(f'{len(vals)} parameters written by the reference to {PKL}; {n} product-written parameters read back by the reference')


if __name__ == '__main__':
    main()
