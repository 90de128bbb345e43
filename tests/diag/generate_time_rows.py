"""generate_data throughput of the row-matrix models at the LHCO shape (N=279, batches of 128), one stream vs the two-stream
pipeline.  Diagnostic.  usage: generate_time_rows.py [lhco_transformer|lhco_crossattention|jetclass]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench_secondary import WORKLOADS
from particle_fm_amd.models import SetFlowMatchingLitModule
from particle_fm_amd.utils.data_generation import generate_data

name = sys.argv[1] if len(sys.argv) > 1 else "lhco_crossattention"
hp, B, n_min, C, flop, what = WORKLOADS[name]
torch.manual_seed(1)
m = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(hp))
with torch.no_grad():
    for p in m.parameters():
        if float(p.abs().sum()) == 0.0 and p.dim() == 2:
            p.uniform_(-1.0, 1.0).div_(p.shape[1] ** 0.5)
N = hp["num_particles"]
n = 6 * B
gen = torch.Generator().manual_seed(2)
nv = torch.randint(n_min, N + 1, (n,), generator=gen)
mask = (torch.arange(N)[None] < nv[:, None]).float().unsqueeze(-1)
cond = torch.randn(n, C, generator=gen) if C else None
for pipe in (False, True, True, False, True):
    data, dt = generate_data(m, n, cond=cond, batch_size=B, device="cuda", variable_set_sizes=True, mask=mask, verbose=False, pipeline=pipe)
    print(f"{name} pipeline={pipe}: {dt*1e3:.1f} ms for {n - B} timed jets -> {(n - B)/dt:.0f} jets/s")
