"""Secondary measurements (BASELINE configs[2] decode latency, configs[4] SampleRNN sample loop, mu-law
quantiser bandwidth).  Development / documentation aid; the driver's headline bench is bench.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device("cuda:0")
out = {}

# ---- configs[2]: autoregressive decode, batch 16, 1000 frames, MSE head (greedy), hipGraph
from parrot_amd.model import Parrot
m = Parrot(device=dev, num_layers=2, rnn_h_dim=1024, readouts_dim=1024, encoder_type='bidirectional',
           weak_feedback=True, use_graph=True).initialize()
g = torch.Generator().manual_seed(0)
N, U, S = 16, 100, 1000
lab = torch.randint(0, 43, (N, U), generator=g)
lm = torch.ones(N, U)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    outs = m.sample_model_device(lab, lm, None, N, S)
    torch.cuda.synchronize(); dt = time.time() - t0
out["decode_cfg3"] = {"batch": N, "frames": S, "seconds": round(dt, 4), "us_per_step": round(1e6 * dt / S, 2),
                      "frames_per_s": round(N * S / dt, 1)}
m.close()

# ---- configs[4]: SampleRNN 3-tier GRU D=1024, batch 32, greedy, 16 kHz mu-law
from parrot_amd.sampleRNN import lib
from parrot_amd.sampleRNN.models.conditional import three_tier as tt
lib.delete_all_params(); lib.set_device(dev)
tt.configure(DIM=1024, EMB_SIZE=256)
B, T = 32, 26  # 25 generated big frames = 2000 samples per stream
seq = torch.randint(0, 256, (2, 160 + 80), generator=g).to(dev)
with torch.no_grad():  # registers all parameters with the reference initialisation
    tt.compute_cost(seq, torch.randn(2, 2, 63, device=dev), torch.zeros(2, 1, 1024, device=dev),
                    torch.zeros(2, 1, 1024, device=dev), 1, torch.ones(2, 240, device=dev))
gen = tt.DeviceGenerator(B, T, temperature=0.0)
feats = torch.randn(T, B, 63, generator=g).numpy()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    s = gen.generate(feats)
    torch.cuda.synchronize(); dt = time.time() - t0
nsamp = (T - 1) * 80
out["samplernn_cfg5"] = {"batch": B, "samples_per_stream": nsamp, "seconds": round(dt, 4),
                         "us_per_sample_step": round(1e6 * dt / nsamp, 2),
                         "samples_per_s": round(B * nsamp / dt, 1),
                         "x_realtime_per_stream": round(nsamp / dt / 16000.0, 3)}
gen.close()

# ---- mu-law quantiser: x ~ N(0,1) [32, 16000*8]
from parrot_amd import ops
x = torch.randn(32, 128000, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        q = ops.batch_quantize(x, 256, "mu-law")
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
out["mulaw"] = {"elements": x.numel(), "us": round(1e6 * dt, 1), "GBps_alg": round(x.numel() * 6 / dt * 1e-9, 1)}
print(json.dumps(out))
