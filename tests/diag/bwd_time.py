"""Diagnostic: the three backward launches (chain, dW GEMM, reduce) of the jet-resident EPiC loss on the bench batch, HIP events.
    [PFM_DIAG=1 PFM_LIB_PATH=tests/diag/libtr_X.so] python tests/diag/bwd_time.py [B] [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd import hip_ops
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd.models import SetFlowMatchingLitModule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
net = model.flows[0].net
lay = net.layout(150)
blob = net.packed_weights(150)
x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(B, 150, 3, 12345))
maskf = mask.reshape(B, -1).float().contiguous()
t = torch.rand(B, device=dev)
z = torch.randn_like(x)
parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
inv = (1.0 / count.sum()).reshape(1)
one = torch.ones(1, device=dev)
gblob = torch.zeros_like(blob)
for _ in range(3):
    hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, gblob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv, one, gblob)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('PFM_LIB_PATH', 'in-tree')}: B={B}: backward (chain + dW + reduce) {e0.elapsed_time(e1) / reps * 1e3:.0f} us")
