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
ex = w[600 + 160:600 + 168].view(np.int64).astype(np.float64) / 100.0  # kernel entry | prologue done | steps done | tail done
print("prologue (entry -> first step): %.2f us   tail (next frame's tier input): %.2f us   entry -> tail done: %.2f us"
      % (ex[1] - ex[0], ex[3] - ex[2], ex[3] - ex[0]))
fx = w[600 + 168:600 + 186].view(np.int64).astype(np.float64) / 100.0
if fx[0] > 0:
    b2, b5 = st[5, 2], st[5, 5]
    print("step 5, L3 product (from x1 taken): FMAs %.2f | DPP fold %.2f | barrier %.2f | final sums %.2f | published (stamp 3) %.2f"
          % (fx[0] - b2, fx[1] - fx[0], fx[2] - fx[1], fx[3] - fx[2], st[5, 3] - fx[3]))
    print("step 5, output product (from x2 taken): FMAs %.2f | DPP fold %.2f | barrier %.2f | final sums %.2f | published %.2f | logits taken %.2f"
          % (fx[4] - b5, fx[5] - fx[4], fx[6] - fx[5], fx[7] - fx[6], fx[8] - fx[7], st[5, 6] - fx[8]))
# end-to-end: a longer utterance through the graph path
T2 = int(os.environ.get("SR_T", "25"))
gen2 = tt.DeviceGenerator(B, T2, temperature=0.0)
feats2 = np.random.RandomState(1).randn(T2, B, 63).astype('float32')
gen2.generate(feats2); torch.cuda.synchronize()
import time
ts = []
for _ in range(7):
    t = time.perf_counter(); out = gen2.generate(feats2); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
ns = (T2 - 1) * 80  # the first big frame is the zero prefix: T2 - 1 periods are generated
print("generation, %d samples x %d streams: %.2f us per sample step (min of 7; median %.2f)"
      % (ns, B, min(ts) * 1e6 / ns, float(np.median(ts)) * 1e6 / ns))
