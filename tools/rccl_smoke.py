"""RCCL on this box: a one-rank process group (backend "nccl" = RCCL on ROCm), a sum all-reduce of a bucket of the
configs[3] gradient size, a broadcast and a barrier on cuda:0.  It does not measure scaling (one GPU) -- it shows that the
library initialises and runs collectives in this image with the environment the multi-GPU launch uses."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
n = 69_450_000  # f32 words of the configs[3] flat gradient (277.8 MB)
g = torch.ones(n, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_reduce(g)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"all_reduce of {4 * n / 1e6:.1f} MB, world 1: {1e3 * dt:.3f} ms")
dist.broadcast(g, src=0)
dist.barrier()
assert float(g[0]) == 1.0 and float(g[-1]) == 1.0
print("backend", dist.get_backend(), "nccl version", torch.cuda.nccl.version(), "HSA_ENABLE_IPC_MODE_LEGACY",
      os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
dist.destroy_process_group()
