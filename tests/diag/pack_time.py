"""Diagnostic: the midpoint sampler on batches of equal-size jets, packed (two jets per workgroup where they fit) vs PFM_PACK=0.
    python tests/diag/pack_time.py            (run twice: with and without PFM_PACK=0 in the environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd.models import SetFlowMatchingLitModule

dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
N, F = 150, 3
for B, n in ((256, 32), (256, 64), (512, 64), (256, 150), (512, 32)):
    mask = (torch.arange(N)[None] < torch.full((B, 1), n)).float().unsqueeze(-1).to(dev)
    z = torch.randn(B, N, F).to(dev) * mask
    with torch.no_grad():
        for _ in range(2):
            model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100)
        e1.record()
        torch.cuda.synchronize()
    print(f"B={B} n={n}: {e0.elapsed_time(e1)/3:.2f} ms per 100-step sample  (PFM_PACK={os.environ.get('PFM_PACK')})")
