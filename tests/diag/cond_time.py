"""Diagnostic: the midpoint sampler on CONDITIONED jets (2 global + 2 local values, N = 150, 6 layers), lean evaluation with the
cond table vs the generic kernel (PFM_F_GENERIC_SAMPLER)."""
import sys, time
sys.path.insert(0, ".")
import torch
from particle_fm_amd.layout import EpicConfig, EpicLayout
from particle_fm_amd import hip_ops
cfg = EpicConfig(num_particles=150, features=3, hidden_dim=128, latent=10, layers=6, frequencies=16, t_local_cat=True, t_global_cat=True,
                 global_cond_dim=2, local_cond_dim=2)
gen = torch.Generator().manual_seed(0)
state = {}
for name, i, o in cfg.linear_shapes():
    state[name + ".weight_v"] = torch.randn(o, i, generator=gen) / i ** 0.5
    state[name + ".weight_g"] = torch.rand(o, 1, generator=gen) + 0.5
    state[name + ".bias"] = torch.randn(o, generator=gen) * 0.1
res = {}
for B, lo, hi in ((256, 32, 32), (256, 150, 150), (256, 30, 150), (1024, 30, 150)):
    n = torch.randint(lo, hi + 1, (B,), generator=gen)
    mask = (torch.arange(150)[None] < n[:, None]).float().unsqueeze(-1).cuda()
    z = torch.randn(B, 150, 3, generator=gen).cuda()
    cond = torch.randn(B, 2, generator=gen).cuda()
    for flags, name in ((1, "lean"), (1 | 32, "generic")):
        lay = EpicLayout(cfg, flags=flags)
        blob = lay.pack_blob(state).cuda()
        for _ in range(2):
            out = hip_ops.epic_sample_midpoint(lay, blob, z, cond, mask, ode_steps=100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = hip_ops.epic_sample_midpoint(lay, blob, z, cond, mask, ode_steps=100)
        torch.cuda.synchronize()
        res[name] = out
        print(f"B={B} n={lo}..{hi} {name:8s}: {(time.perf_counter()-t0)/3*1e3:8.2f} ms", flush=True)
    print("   max |lean - generic| =", float((res["lean"] - res["generic"]).abs().max()))
