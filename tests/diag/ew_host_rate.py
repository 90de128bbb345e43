"""Diagnostic: host enqueue time against GPU time of one cfg-5 midpoint sample (is the sampler bound by the host's launch rate?).
   python tests/diag/ew_host_rate.py [jets] [ode_steps]"""
import copy
import sys
import time

import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench_secondary import WORKLOADS, make_batch  # noqa: E402
from particle_fm_amd.models import SetFlowMatchingLitModule  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hp, _, n_min, C, flop, what = WORKLOADS["jetclass"]
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(hp)).to(dev)
N, F = hp["num_particles"], hp["features"]
x, mask, cond = (a.to(dev) for a in make_batch(B, N, F, C, n_min, 12345))
z = (torch.randn(B, N, F) * mask.cpu()).to(dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s), torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = model(z, cond=cond, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        print(f"call {it}: host enqueue {1e3 * (t1 - t0):.1f} ms, until the GPU is done {1e3 * (t2 - t0):.1f} ms", flush=True)
