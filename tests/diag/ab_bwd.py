"""Diagnostic: A/B of two library builds on the backward of the jet-resident EPiC loss -- the gradient blob of uniform batches (every jet n
valid particles, n = 1 .. N) and of the bench mix must agree (bit for bit if the builds differ in scheduling only; an FMA contracted differently shows up at 1e-8).
    python tests/diag/ab_bwd.py libA.so libB.so      (each run in a child process: one library per process)"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

if len(sys.argv) == 3:
    outs = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ, PFM_LIB_PATH=lib, PFM_DIAG="1", PFM_AB_OUT=f"/tmp/ab_bwd_{len(outs)}.npz")
        outs.append(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True))
        if outs[-1].returncode:
            print(outs[-1].stdout, outs[-1].stderr)
            sys.exit(1)
    import numpy as np
    a, b = np.load("/tmp/ab_bwd_0.npz"), np.load("/tmp/ab_bwd_1.npz")
    worst = {}
    for k in a.files:
        d = np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-30)
        if d > 0:
            worst[k] = float(d)
    big = {k: v for k, v in worst.items() if v > 1e-5}
    print("cases", len(a.files), "bitwise different", len(worst), "max rel (to the largest gradient entry)", max(worst.values()) if worst else 0.0)
    print("beyond 1e-5:", big)
    sys.exit(1 if big else 0)

import hashlib
import torch
import bench
from particle_fm_amd import hip_ops
from particle_fm_amd.models import SetFlowMatchingLitModule

dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
net = model.flows[0].net
N = 150
lay = net.layout(N)
blob = net.packed_weights(N)


RES = {}


def run(maskf, tag):
    B = maskf.shape[0]
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.randn(B, N, 3, generator=g).to(dev)) * maskf[..., None]
    t = torch.rand(B, generator=g).to(dev)
    z = torch.randn(B, N, 3, generator=g).to(dev)
    parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
    inv = (1.0 / count.sum()).reshape(1)
    one = torch.ones(1, device=dev)
    gblob = torch.zeros_like(blob)
    hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, gblob)
    torch.cuda.synchronize()
    # only the slots that are gradients of parameters (the blob's padding rows receive products with never-written LDS padding)
    gp = torch.as_tensor(lay.src_gpos, device=dev).long()
    RES[tag] = gblob[gp[gp >= 0]].cpu().numpy()


for n in list(range(1, 40)) + list(range(40, N + 1, 7)) + [N]:
    run((torch.arange(N)[None] < torch.full((8, 1), n)).float().to(dev).contiguous(), f"n={n}")
x, mask, cond = (a_.to(dev) for a_ in bench.synthetic_batch(64, N, 3, 12345))
run(mask.reshape(64, -1).float().contiguous(), "mix")
import numpy as np
np.savez(os.environ["PFM_AB_OUT"], **RES)
