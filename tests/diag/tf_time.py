"""Timing of the transformer NFE / sampler at BASELINE cfg 4 (B=128, N=279).  Diagnostic, not a test."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from particle_fm_amd import hip_ops_tf as ops
from particle_fm_amd.layout_tf import TfConfig, TfLayout
from oracle.seeded import seeded_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = TfConfig(num_particles=279, global_cond_dim=5)
lay = TfLayout(cfg, flags=(1 if "x3" in sys.argv else 0) | (4 if "valid" in sys.argv else 0))
st = {k: torch.from_numpy(v) for k, v in seeded_state(dict(cfg.param_shapes()), 1).items()}
blob = lay.pack_blob(st).cuda()
gen = torch.Generator().manual_seed(0)
n = torch.randint(20, 280, (B,), generator=gen)
mask = (torch.arange(279)[None] < n[:, None]).float().cuda()
x = torch.randn(B, 279, 3, generator=gen).cuda()
cond = torch.randn(B, 5, generator=gen).cuda()
t = torch.rand(B, generator=gen).cuda()
for _ in range(3):
    v = ops.tf_forward(lay, blob, t, x, cond, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    v = ops.tf_forward(lay, blob, t, x, cond, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
fl = 1365e6 * B
print(f"NFE: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s algorithmic ({B} jets)")
if steps > 1:
    xs = ops.tf_sample_midpoint(lay, blob, x, cond, mask, ode_steps=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xs = ops.tf_sample_midpoint(lay, blob, x, cond, mask, ode_steps=steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sample {steps} steps: {dt*1e3:.1f} ms  {B/dt:.1f} jets/s  {fl*2*(steps-1)/dt/1e12:.1f} TFLOP/s  (host enqueue {t_host*1e3:.1f} ms)")
if len(sys.argv) > 3 and sys.argv[3] == "train":
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    flat = torch.cat([st[k].reshape(-1) for k in lay.keys()]).cuda().requires_grad_(True)
    z = torch.randn(B, 279, 3, generator=gen).cuda()
    for _ in range(2):
        loss = tf_fm_loss(lay, flat, x, t, z, cond, mask.unsqueeze(-1)); loss.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        flat.grad = None
        loss = tf_fm_loss(lay, flat, x, t, z, cond, mask.unsqueeze(-1)); loss.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"loss fwd+bwd: {dt*1e3:.2f} ms  {B/dt:.0f} jets/s  {3*fl/dt/1e12:.1f} TFLOP/s algorithmic")
