"""Diagnostic: do two sampler launches in flight slow each other down through the L2 (two weight snapshots = 2 x ~1.9 MB of hot weights
per XCD L2 of 4 MB)?  Times K back-to-back 100-step samples of the bench batch: one stream; two streams with the SAME blob; two streams
with two different blobs (what bench.py's pipeline does).    python tests/diag/overlap_l2.py [K=8]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd.engine import FusedFMTrainer
from particle_fm_amd.models import SetFlowMatchingLitModule
from particle_fm_amd.utils.streams import concurrent_streams

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
tr = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(256, 150, 3, 12345))
z = (torch.randn(256, 150, 3, generator=torch.Generator().manual_seed(9999)) * mask.cpu()).to(dev)
if os.environ.get('PFM_PACK', '1') == '1':
    model.flows[0].net.set_jet_packing(True)  # (what bench.py does by default)
blobs = [tr.snapshot_blob(150) for _ in range(2)]
streams = concurrent_streams(4, dev)

def run(nstreams, nblobs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(K):
            with torch.cuda.stream(streams[i % nstreams]):
                model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100, weights=blobs[i % nblobs])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K

for _ in range(2):
    run(2, 2)
CASES = (("one stream", 1, 1),) if os.environ.get("PFM_ONLY_ONE") else None
for name, ns, nb in CASES or (("one stream", 1, 1), ("two streams, one blob", 2, 1), ("two streams, two blobs", 2, 2), ("two streams, one blob", 2, 1), ("two streams, two blobs", 2, 2), ("three streams", 3, 2), ("four streams", 4, 2), ("three streams", 3, 2)):
    print(f"{name:26s} {run(ns, nb):8.3f} ms per sample of 256 jets", flush=True)
