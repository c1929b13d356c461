import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARROT_SCHEDULE"] = "4"
import torch
from oracle import parrot_ref as R
from parrot_amd.model import Parrot
from parrot_amd import _lib
from tests.util import make_batch
dev = torch.device("cuda:0")
SMALL = dict(rnn_h_dim=64, readouts_dim=48, encoder_dim=16, input_dim=24, speaker_dim=8, num_speakers=5, encoder_type='bidirectional')
def words(iv, tag):
    c = collections.Counter(iv[:1024].tolist())
    print(f"    {tag}: abort {int(iv[832])} sticky {int(iv[992])} common {c.most_common(3)}", flush=True)
for L, B in ((1, 5), (2, 20), (2, 64), (3, 37), (3, 64)):
    full = dict(SMALL, num_layers=L)
    cfg = R.default_config(**full)
    p = R.init_params(cfg, seed=7, scale_by_fan_in=True)
    m = Parrot(device=dev, use_graph=True, **full).allocate()
    m.set_parameter_values(p)
    T, U = 9, 11
    feat, fm, lab, lm, spk = make_batch(cfg, T, B, U, seed=3, ragged=True)
    print(L, B, flush=True)
    for rep in range(2):
        m.zero_grad()
        ws = m._train_ws.get(('dec', T, B, U))
        if ws is not None:
            words(ws['persist_ws'].view(torch.int32), f"rep {rep} before fwd")
        try:
            cost, upd, av, _ = m.compute_cost(feat.float().to(dev), fm.float().to(dev), lab.to(dev), lm.float().to(dev), None, 1, B)
            ws = m._train_ws.get(('dec', T, B, U))
            torch.cuda.synchronize()
            words(ws['persist_ws'].view(torch.int32), f"rep {rep} after fwd ")
            cost.backward()
            torch.cuda.synchronize()
            words(ws['persist_ws'].view(torch.int32), f"rep {rep} after bwd ")
        except Exception as e:
            print("  EXC", e, flush=True)
            ws = m._train_ws.get(('dec', T, B, U))
            words(ws['persist_ws'].view(torch.int32), f"rep {rep} at failure")
            break
    m.close()
