"""Diagnostic: the train step of bench.py alone (FusedFMTrainer.step on the bench batch), timed with HIP events.
    python tests/diag/train_time.py [B] [reps]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd.engine import FusedFMTrainer
from particle_fm_amd.models import SetFlowMatchingLitModule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
tr = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(B, 150, 3, 12345))
for _ in range(5):
    tr.step((x, mask, cond))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    tr.step((x, mask, cond))
e1.record()
torch.cuda.synchronize()
print(f"B={B}: train step {e0.elapsed_time(e1) / reps:.3f} ms")
