"""ad-hoc timing helper (not a test): python tests/quick_time.py [B] [steps]"""
import sys, time, torch
sys.path.insert(0, ".")  # run from the repository root
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd import hip_ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = load_golden("jetnet150")
lay = EpicLayout(cfg_of(g.hp), flags=flags)
blob = lay.pack_blob(g.state, "flows.0.net.").cuda()
gen = torch.Generator().manual_seed(0)
n = torch.randint(30, 151, (B,), generator=gen)
mask = (torch.arange(150)[None] < n[:, None]).float().unsqueeze(-1).cuda()
x = torch.randn(B, 150, 3, generator=gen).cuda() * mask
t = torch.rand(B, generator=gen).cuda()
for name, fn in [("forward", lambda: hip_ops.epic_forward(lay, blob, t, x, None, mask)),
                 ("sample", lambda: hip_ops.epic_sample_midpoint(lay, blob, x, None, mask, ode_steps=steps))]:
    fn(); torch.cuda.synchronize()
    reps = 20 if name == "forward" else 2
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nfe = 1 if name == "forward" else 2 * (steps - 1)
    print(f"{name}: B={B} flags={flags} {dt*1e3:.3f} ms  -> {B/dt:.1f} jets/s, {B*nfe*84.22e6/dt/1e12:.2f} TFLOP/s algorithmic")
