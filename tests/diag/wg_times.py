"""Diagnostic: per-workgroup start / end times of the lean sampler inside a pipeline of launches (library with -DPFM_WGT:
    hipcc ... -DPFM_WGT -c particle_fm_amd/csrc/epic_kernels.hip -o /tmp/ek_wgt.o; link with the other build/obj/*.hip.o -> tests/diag/libfull_wgt.so
    PFM_DIAG=1 PFM_LIB_PATH=tests/diag/libfull_wgt.so python tests/diag/wg_times.py [K=12] [streams=2]
Prints the period per launch, the sum of workgroup durations / 256 CUs per launch (CU time really spent) and the same for ONE launch alone:
if the pipelined sum equals the lone one, the gap to the period is idle CUs; if it is larger, workgroups slow each other down."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from particle_fm_amd import _lib
from particle_fm_amd.engine import FusedFMTrainer
from particle_fm_amd.models import SetFlowMatchingLitModule
from particle_fm_amd.utils.streams import concurrent_streams

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = SetFlowMatchingLitModule(optimizer=None, **bench.HP).to(dev)
tr = FusedFMTrainer(model, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
x, mask, cond = (a.to(dev) for a in bench.synthetic_batch(256, 150, 3, 12345))
z = (torch.randn(256, 150, 3, generator=torch.Generator().manual_seed(9999)) * mask.cpu()).to(dev)
model.flows[0].net.set_jet_packing(True)
blobs = [tr.snapshot_blob(150) for _ in range(2)]
streams = concurrent_streams(4, dev)
lib = _lib.load()
CAP = 32768
buf = (ctypes.c_ulonglong * (2 * CAP))()
n = ctypes.c_int(0)

def run(k, ns):
    lib.pfm_diag_read_wgt(buf, CAP, ctypes.byref(n))  # reset
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(k):
            with torch.cuda.stream(streams[i % ns]):
                model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100, weights=blobs[i % 2])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    lib.pfm_diag_read_wgt(buf, CAP, ctypes.byref(n))
    m = min(n.value, CAP)
    st = [buf[2 * i] for i in range(m)]
    en = [buf[2 * i + 1] for i in range(m)]
    dur = [(b - a) / 100e3 for a, b in zip(st, en)]  # ms (100 MHz)
    span = (max(en) - min(st)) / 100e3
    return wall, m, sum(dur), span, dur

for _ in range(2):
    run(2, 2)
wall, m, tot, span, dur = run(1, 1)
print(f"one launch alone: {m} workgroups, wall {wall:.2f} ms, span {span:.2f} ms, sum of workgroup durations / 256 = {tot / 256:.3f} ms, longest {max(dur):.2f}")
wall, m, tot, span, dur = run(K, NS)
print(f"{K} launches on {NS} streams: {m} workgroups, wall {wall / K:.3f} ms per launch, sum of workgroup durations / 256 = {tot / 256 / K:.3f} ms per launch "
      f"(CU time really spent), span {span / K:.3f}; idle share {1 - tot / 256 / span:.3f}")

# busy-CU count over time (every workgroup holds one CU: 160 KB of LDS), sampled every 0.25 ms over the steady part of the run
lib.pfm_diag_read_wgt(buf, CAP, ctypes.byref(n))
torch.cuda.synchronize()
with torch.no_grad():
    for i in range(K):
        with torch.cuda.stream(streams[i % NS]):
            model(z, cond=None, mask=mask, reverse=True, ode_solver="midpoint", ode_steps=100, weights=blobs[i % 2])
torch.cuda.synchronize()
lib.pfm_diag_read_wgt(buf, CAP, ctypes.byref(n))
m = min(n.value, CAP)
iv = sorted((buf[2 * i], buf[2 * i + 1]) for i in range(m))
t0 = min(a for a, _ in iv)
T1 = max(b for _, b in iv)
step = 25_000  # 0.25 ms at 100 MHz
lo, hi = t0 + (T1 - t0) // 4, t0 + 3 * (T1 - t0) // 4
counts = []
t = lo
while t < hi:
    counts.append(sum(1 for a, b in iv if a <= t < b))
    t += step
import collections
hist = collections.Counter((c // 8) * 8 for c in counts)
print("busy CUs (middle half of the run), share of samples per bucket of 8:")
for k_ in sorted(hist):
    print(f"  {k_:3d}-{k_ + 7:3d}: {hist[k_] / len(counts):6.3f}")
print("mean busy", sum(counts) / len(counts))
# gaps: for every workgroup start, how long had the longest-idle CU been free?  (approximation: time since the (busy-256)th last end)
line = " ".join(f"{c:3d}" for c in counts[:160])
print("first 40 ms of the window, one sample per 0.25 ms:\n" + line)
