"""Sampler throughput of the jet-resident EPiC kernel, fp32 vs bf16 MFMA operands (BASELINE cfg 2 and cfg 3 shapes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from particle_fm_amd.models import SetFlowMatchingLitModule

for N, B in ((30, 1024), (150, 256), (150, 1024)):
    torch.manual_seed(12345)
    m = SetFlowMatchingLitModule(optimizer=None, model="epic", features=3, hidden_dim=128, num_particles=N, frequencies=16, layers=6,
                                 latent=10, t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="cosine").cuda()
    gen = torch.Generator().manual_seed(0)
    n = torch.randint(max(10, N // 5), N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1).cuda()
    z = torch.randn(B, N, 3, generator=gen).cuda() * mask
    for prec in ("fp32", "f16x3", "bf16"):
        m.flows[0].net.set_precision(prec)
        with torch.no_grad():
            for _ in range(2):
                out = m(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = m(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"N={N} B={B} {prec}: {dt*1e3:.2f} ms  {B/dt:.0f} jets/s")
