"""Diagnostic: dump the backward scratch (gradient rows da, per-jet records) of one uniform batch for two library builds and say where they
differ.   python tests/diag/ab_bwd_dump.py libA.so libB.so n"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

if sys.argv[1] != "child":
    n = sys.argv[3]
    for i, lib in enumerate(sys.argv[1:3]):
        env = dict(os.environ, PFM_LIB_PATH=lib, PFM_DIAG="1")
        r = subprocess.run([sys.executable, __file__, "child", n, f"/tmp/abdump{i}.npy"], env=env, capture_output=True, text=True)
        print(r.stdout[-2000:], r.stderr[-2000:])
    a, b = np.load("/tmp/abdump0.npy"), np.load("/tmp/abdump1.npy")
    meta = np.load("/tmp/abdump0.npy.meta.npy")
    rec0, da0, part0, nblk, N, recf = (int(v) for v in meta)
    B = 2
    da_a = a[da0:da0 + B * nblk * N * 128].reshape(B, nblk, N, 128)
    da_b = b[da0:da0 + B * nblk * N * 128].reshape(B, nblk, N, 128)
    nn = int(n)
    for blk in range(nblk):
        d = np.abs(da_a[0, blk, :nn] - da_b[0, blk, :nn])
        rows = np.nonzero(d.max(axis=1) > 0)[0]
        cols = np.nonzero(d.max(axis=0) > 0)[0]
        print("da block", blk, "max diff", d.max(), "rows", rows[:20], "cols", cols[:20], "ref max", np.abs(da_a[0, blk, :nn]).max())
    d12a, d12b = da_a[0, nblk - 1, :nn], da_b[0, nblk - 1, :nn]
    idx = np.argwhere(d12a != d12b)[:24]
    for r, c in idx:
        print("blk", nblk - 1, "row", r, "col", c, "old", d12a[r, c], "new", d12b[r, c], "ratio", d12b[r, c] / d12a[r, c])
    ra = a[rec0:rec0 + B * recf].reshape(B, recf)
    rb = b[rec0:rec0 + B * recf].reshape(B, recf)
    d = np.abs(ra[0] - rb[0])
    names = [("VIN", 0, 352), ("VIN2", 352, 208), ("DAG1", 560, 128), ("DAG2", 688, 16), ("DBJ1", 704, 128), ("DBJ2", 832, 128), ("GOUT", 960, 16)]
    for st in range(7):
        for nm, o, ln in names:
            seg = d[st * 976 + o: st * 976 + o + ln]
            if nm in ("VIN", "VIN2"):
                seg = seg[:298 if nm == "VIN" else 160]
            if seg.max() > 0:
                print("rec stage", st, nm, "max diff", seg.max(), "at", np.nonzero(seg > 0)[0][:16], "ref max", np.abs(ra[0][st * 976 + o: st * 976 + o + ln]).max())
    tail = d[7 * 976:]
    print("rec head part (dW3 | dWx | db3) max diff", tail.max(), np.nonzero(tail > 0)[0][:16])
    sys.exit(0)

import ctypes
import torch
import bench
from particle_fm_amd import _lib, hip_ops
from particle_fm_amd.models import SetFlowMatchingLitModule

n, out = int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
net = model.flows[0].net
N = 150
lay = net.layout(N)
blob = net.packed_weights(N)
B = 2
maskf = (torch.arange(N)[None] < torch.full((B, 1), n)).float().to(dev).contiguous()
g = torch.Generator(device="cpu").manual_seed(7)
x = (torch.randn(B, N, 3, generator=g).to(dev)) * maskf[..., None]
t = torch.rand(B, generator=g).to(dev)
z = torch.randn(B, N, 3, generator=g).to(dev)
parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
inv = (1.0 / count.sum()).reshape(1)
one = torch.ones(1, device=dev)
gblob = torch.zeros_like(blob)
scr = hip_ops.epic_backward_scratch(lay, B, dev)
scr.zero_()
hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, gblob)
torch.cuda.synchronize()
np.save(out, scr.cpu().numpy())
lib = _lib.load()
layers = 6
recf = (layers + 1) * (352 + 208 + 128 + 16 + 128 + 128 + 16) + 16 * 128 * 2 + 16
nrows_f = (B + 63) & ~63
np.save(out + ".meta.npy", np.array([nrows_f, nrows_f + B * recf, 0, 2 * layers + 1, N, recf]))
print("scratch floats", scr.numel())
