"""Phase timers of the persistent SampleRNN sample kernel (PARROT_SR_TIMING=1): workgroup 0 of team 0 stamps, per step,
0 step start | 1 x1 published (part + newest row, no product) | 2 x1 taken | 3 L3 product done, x2 published |
4 next step's part summed | 5 x2 taken | 6 logits taken | 7 pick done.  Prints the median interval of each stage over the
last launch's steps."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARROT_SR_TIMING"] = "1"
from parrot_amd.sampleRNN import lib
from parrot_amd.sampleRNN.models.conditional import three_tier as tt

dev = torch.device("cuda:0")
lib.delete_all_params(); lib.set_device(dev)
tt.configure(DIM=1024, EMB_SIZE=256, RNN_TYPE='GRU', N_RNN=1)
tt.init_random_params(seed=3) if hasattr(tt, 'init_random_params') else None
from oracle import samplernn_ref as S
lib.set_params(S.init_params(S.config(), seed=5, perturb=0.2))
B, T = 32, 6
gen = tt.DeviceGenerator(B, T, temperature=0.0, use_graph=False)
feats = np.random.RandomState(0).randn(T, B, 63).astype('float32')
gen.generate(feats); torch.cuda.synchronize()
w = gen.ws['persist_ws'][:1024].cpu().view(torch.int32).numpy()
st = w[600:600 + 160].view(np.int64).reshape(10, 8).astype(np.float64) / 100.0  # us
names = ["publish x1 (part + newest row)", "take x1", "L3 product + publish x2", "next step's part (gather-sum)",
         "take x2", "output product + publish logits, take logits", "pick"]
d = np.diff(st, axis=1)
for q, n in enumerate(names):
    print(f"{n:55s} median {np.median(d[1:9, q]):6.2f} us   (min {d[1:9, q].min():.2f}, max {d[1:9, q].max():.2f})")
print(f"{'step total':55s} median {np.median(st[2:9, 0] - st[1:8, 0]):6.2f} us")
print("launch span (10 steps, without prologue): %.2f us" % (st[9, 7] - st[0, 0]))
