"""Freezes outputs of the ORACLE (oracle/parrot_ref.py, fp64) for small Parrot configurations into
tests/golden/parrot_golden.npz.  These vectors are oracle-generated, NOT reference-generated (Theano/Blocks cannot
run here; parity of this path is unpinned, see oracle/__init__.py): they guard against silent drift of the oracle
and give the HIP path a fixed target that does not depend on the oracle code of the day.

    python tests/golden/make_parrot_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import parrot_ref as R  # noqa: E402
from tests.util import make_batch  # noqa: E402

CASES = {
    'gru3_fb_spk': dict(num_layers=3, full_feedback=True, use_speaker=True),
    'gru2_gmm': dict(num_layers=2, weak_feedback=True, which_cost='GMM', k_gmm=3),
    'lstm2': dict(num_layers=2, weak_feedback=True, cell_type='lstm'),
    'gru2_ln': dict(num_layers=2, weak_feedback=True, layer_norm=True),
}
SMALL = dict(rnn_h_dim=32, readouts_dim=32, encoder_dim=16, input_dim=24, speaker_dim=8, num_speakers=5,
             output_dim=7, encoder_type='bidirectional')
GRAD_KEYS = ['/parrot/rnn1.initial_state', '/parrot/h1_to_att/fork_kappa.W', '/parrot/att_to_readout.W',
             '/parrot/inp_to_h1/fork_rnn1_inputs.W', '/parrot/encoder/embed_label.W']


def run_case(kw):
    cfg = R.default_config(**dict(SMALL, **kw))
    p = R.init_params(cfg, seed=11, scale_by_fan_in=True)
    feat, fm, lab, lm, spk = make_batch(cfg, 6, 4, 9, seed=21, ragged=True, speaker=cfg['use_speaker'])
    for v in p.values():
        v.requires_grad_()
    cost, carry, av, ex = R.compute_cost(p, cfg, feat, fm, lab, lm, spk, 1)
    cost.backward()
    out = dict(cost=cost.detach().numpy(), next_x=av[0].detach().numpy(), kappa=av[1].detach().numpy(),
               w=av[2].detach().numpy(), phi=av[4].detach().numpy())
    for k in GRAD_KEYS:
        out['grad:' + k] = p[k].grad.numpy()
    with torch.no_grad():
        S = 5
        g = torch.Generator().manual_seed(5)
        unif = torch.rand(S, 4, generator=g, dtype=torch.float64)
        noise = torch.randn(S, 4, cfg['output_dim'], generator=g, dtype=torch.float64)
        sm = R.sample_model({k: v.detach() for k, v in p.items()}, cfg, lab, lm, spk, S, unif=unif, noise=noise)
    out['sample_x'] = sm[0].numpy()
    out['sample_k'] = sm[1].numpy()
    return out


if __name__ == '__main__':
    blob = {}
    for name, kw in CASES.items():
        for k, v in run_case(kw).items():
            blob[f'{name}|{k}'] = v
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'parrot_golden.npz')
    np.savez_compressed(path, **blob)
    print('wrote', path, len(blob), 'arrays', os.path.getsize(path), 'bytes')
