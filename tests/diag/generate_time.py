"""generate_data throughput at the JetNet-150 shape (2560 jets in batches of 256), one stream vs two.  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from particle_fm_amd.models import SetFlowMatchingLitModule
from particle_fm_amd.utils.data_generation import generate_data

HP = dict(model="epic", features=3, hidden_dim=128, num_particles=150, frequencies=16, layers=6, latent=10, activation="leaky_relu",
          wrapper_func="weight_norm", t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="cosine", loss_type="FM-OT")
torch.manual_seed(1)
m = SetFlowMatchingLitModule(optimizer=None, **HP)
n = 2560
gen = torch.Generator().manual_seed(2)
nv = torch.randint(30, 151, (n,), generator=gen)
mask = (torch.arange(150)[None] < nv[:, None]).float().unsqueeze(-1)
for pipe in (False, True, True, True, False, True, True):
    data, dt = generate_data(m, n, batch_size=256, device="cuda", variable_set_sizes=True, mask=mask, verbose=False, pipeline=pipe)
    print(f"pipeline={pipe}: {dt*1e3:.1f} ms for {n - 256} timed jets -> {(n - 256)/dt:.0f} jets/s")
