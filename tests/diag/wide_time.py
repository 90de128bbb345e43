"""Timing of the wide EPiC NFE / sampler at BASELINE cfg 5 (B=256, N=128, H=300, 20 layers).  Diagnostic."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from particle_fm_amd import hip_ops_wide as ops
from particle_fm_amd.layout import EpicConfig
from particle_fm_amd.layout_wide import EpicWideLayout
from oracle.seeded import seeded_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = EpicConfig(num_particles=128, features=13, hidden_dim=300, latent=16, layers=20, frequencies=16, t_local_cat=True,
                 t_global_cat=True, global_cond_dim=12, local_cond_dim=0)
lay = EpicWideLayout(cfg, with_backward=False, flags=1 if "x3" in sys.argv else 0)
shapes = {}
for name, i, o in cfg.linear_shapes():
    shapes[name + ".bias"] = (o,); shapes[name + ".weight_g"] = (o, 1); shapes[name + ".weight_v"] = (o, i)
st = {k: torch.from_numpy(v) for k, v in seeded_state(shapes, 1).items()}
blob = lay.pack_blob(st).cuda()
print("blob MB", blob.numel() * 4 / 1e6)
gen = torch.Generator().manual_seed(0)
n = torch.randint(20, 129, (B,), generator=gen)
mask = (torch.arange(128)[None] < n[:, None]).float().cuda()
x = torch.randn(B, 128, 13, generator=gen).cuda() * mask[..., None]
cond = torch.randn(B, 12, generator=gen).cuda()
t = torch.rand(B, generator=gen).cuda()
for _ in range(3):
    v = ops.ew_forward(lay, blob, t, x, cond, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    v = ops.ew_forward(lay, blob, t, x, cond, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
fl = 1083.1e6 * B
print(f"NFE: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s algorithmic ({B} jets)")
if steps > 1:
    xs = ops.ew_sample_midpoint(lay, blob, x, cond, mask, ode_steps=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xs = ops.ew_sample_midpoint(lay, blob, x, cond, mask, ode_steps=steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sample {steps} steps: {dt*1e3:.1f} ms  {B/dt:.1f} jets/s  {fl*2*(steps-1)/dt/1e12:.1f} TFLOP/s  (host enqueue {t_host*1e3:.1f} ms)")
if len(sys.argv) > 3 and sys.argv[3] == "train":
    from particle_fm_amd.fm_loss_wide import epic_wide_fm_loss
    layb = EpicWideLayout(cfg)
    stg = {k: v.cuda().requires_grad_(True) for k, v in st.items()}
    z = torch.randn(B, 128, 13, generator=gen).cuda()
    def step():
        for v_ in stg.values():
            v_.grad = None
        loss = epic_wide_fm_loss(layb, layb.source_vector(stg), x, t, z, cond, mask.unsqueeze(-1))
        loss.backward()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"loss fwd+bwd (incl. weight-norm autograd): {dt*1e3:.2f} ms  {B/dt:.0f} jets/s  {3*fl/dt/1e12:.1f} TFLOP/s algorithmic")
