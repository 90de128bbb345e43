"""Diagnostic: does a pipeline of lean-sampler launches lose less CU time to the in-order workgroup dispatcher (a launch's workgroups go
round-robin over the 8 XCDs, in order: one that waits for its XCD holds up all behind it) when every 256-jet batch goes out as H calls of
256 / H jets on H streams?     python tests/diag/split_launch.py [K=24] [H=2] [batches in flight=2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd.engine import FusedFMTrainer
from particle_fm_amd.models import SetFlowMatchingLitModule
from particle_fm_amd.utils.streams import concurrent_streams

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
tr = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(256, 150, 3, 12345))
z = (torch.randn(256, 150, 3, generator=torch.Generator().manual_seed(9999)) * mask.cpu()).to(dev)
model.flows[0].net.set_jet_packing(True)
blobs = [tr.snapshot_blob(150) for _ in range(2)]
streams = concurrent_streams(8, dev)
# parts with the same multiplicity distribution: jets sorted by length, dealt round
order = torch.argsort(mask.sum((1, 2)).cpu(), descending=True)
def parts_of(h):
    pp = [order[q::h].to(dev) for q in range(h)]
    return [z[p].contiguous() for p in pp], [mask[p].contiguous() for p in pp]
PARTS = {h: parts_of(h) for h in (2, 4, 8)}

def run(k, h, d):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(k):
            for q in range(h):
                with torch.cuda.stream(streams[(i % d) * h + q]):
                    if h == 1:
                        model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100, weights=blobs[i % 2])
                    else:
                        model(PARTS[h][0][q], cond=None, mask=PARTS[h][1][q], reverse=True, ode_solver="midpoint", ode_steps=100, weights=blobs[i % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k

for _ in range(2):
    run(2, H, D)
for h, d in ((1, 2), (H, D), (1, 2), (H, D), (4, 2), (2, 3), (4, 1), (8, 1)):
    if h * d <= 8:
        print(f"{h} call(s) per 256-jet batch, {d} batches in flight: {run(K, h, d):8.3f} ms per batch", flush=True)
