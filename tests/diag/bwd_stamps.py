"""Diagnostic: where workgroup 0 of the backward chain kernel spends its cycles (s_memtime stamps; library built in the CPU container by
tests/diag/build_bwd_stamps.sh: epic_train.hip under -DPFM_BDIAG).   PFM_DIAG=1 PFM_LIB_PATH=tests/diag/libtr_stamps.so python tests/diag/bwd_stamps.py [n_particles]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd import _lib, hip_ops
from particle_fm_amd.models import SetFlowMatchingLitModule

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = 256
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
net = model.flows[0].net
lay = net.layout(150)
blob = net.packed_weights(150)
maskf = (torch.arange(150)[None] < torch.full((B, 1), n)).float().to(dev).contiguous()
x = torch.randn(B, 150, 3, device=dev) * maskf[..., None]
t = torch.rand(B, device=dev)
z = torch.randn_like(x)
parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
inv = (1.0 / count.sum()).reshape(1)
one = torch.ones(1, device=dev)
gblob = torch.zeros_like(blob)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 1024)()
cnt = ctypes.c_int(0)
for it in range(3):
    hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, gblob)
    lib.pfm_diag_read_bwd_stamps(buf, ctypes.byref(cnt))
names = {0: "start", 1: "setup done", 2: "head done", 10: "layer start", 11: "(1) done", 12: "(3) gemm_dx done", 13: "vin/tgemv/rec done",
         14: "global_backward done", 20: "layers done", 21: "stem global + da2s done", 22: "stem gemm_dx done", 30: "end"}
prev = None
tot = {}
for i in range(cnt.value):
    sid, tk = buf[2 * i], buf[2 * i + 1]
    if prev is not None:
        key = f"{names.get(prev[0], prev[0])} -> {names.get(sid, sid)}"
        tot.setdefault(key, []).append(tk - prev[1])
    prev = (sid, tk)
total = sum(sum(v) for v in tot.values())
print(f"n = {n} particles: {total} cycles per jet")
for k, v in tot.items():
    print(f"{k:50s} n={len(v):2d} mean {sum(v)/len(v):9.0f}  sum {sum(v):8d} ({100*sum(v)/total:4.1f}%)")
