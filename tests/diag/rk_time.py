import sys, os, time
sys.path.insert(0, ".")
import torch
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd import hip_ops
g = load_golden("jetnet150")
gen = torch.Generator().manual_seed(0)
B, N = 256, 150
n = torch.randint(30, 151, (B,), generator=gen)
mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1).cuda()
z = torch.randn(B, N, 3, generator=gen).cuda()
for flags, name in ((1, "lean"), (1 | 32, "generic")):
    lay = EpicLayout(cfg_of(g.hp), flags=flags)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    for solver, steps in (("euler", 100), ("rk4", 50), ("midpoint", 100)):
        for _ in range(2):
            hip_ops.epic_sample_rk(lay, blob, z, None, mask, ode_steps=steps, solver=solver)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            hip_ops.epic_sample_rk(lay, blob, z, None, mask, ode_steps=steps, solver=solver)
        torch.cuda.synchronize()
        print(f"{name:8s} {solver:9s} {steps:4d} steps: {(time.perf_counter()-t0)/3*1e3:8.2f} ms", flush=True)
