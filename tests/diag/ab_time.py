"""Diagnostic (not a test, not shipped): A/B timing of sampler builds.  Each argument is a shared library built from
particle_fm_amd/csrc/epic_kernels.hip alone (self-contained translation unit):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iparticle_fm_amd/csrc particle_fm_amd/csrc/epic_kernels.hip -o tests/diag/lib_X.so
    python tests/diag/ab_time.py tests/diag/lib_A.so tests/diag/lib_B.so
Prints ms per 100-step midpoint sample (198 evaluations) for uniform batches and for the bench's multiplicity mix, and the
max deviation between the builds' outputs."""
import ctypes, os, sys
sys.path.insert(0, ".")
import torch
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd.hip_ops import midpoint_grid

P = ctypes.c_void_p
g = load_golden("jetnet150")
hp = dict(g.hp); hp["num_particles"] = 150
FLAGS = int(os.environ.get("PFM_FLAGS", "1"))  # 1 SKIP_MASKED_TAIL | 2 BF16 | 32 GENERIC_SAMPLER ...
lay = EpicLayout(cfg_of(hp), flags=FLAGS)
blob = lay.pack_blob(g.state, "flows.0.net.").cuda()
ts, dts = midpoint_grid(100)
ts, dts = ts.cuda(), dts.cuda()
NI = 99
gen = torch.Generator().manual_seed(0)
cases = []
for Bn, n in ((256, 32), (256, 64), (256, 96), (256, 150), (512, 48)):
    mask = (torch.arange(150)[None] < torch.full((Bn, 1), n)).float()
    cases.append((f"B={Bn} n={n}", Bn, mask))
for Bn in (256, 1024):
    nn = torch.randint(30, 151, (Bn,), generator=gen)
    cases.append((f"B={Bn} U{{30..150}}", Bn, (torch.arange(150)[None] < nn[:, None]).float()))
outs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.pfm_epic_sample_scratch_floats.restype = ctypes.c_int64
    for name, Bn, mask in cases:
        maskd = mask.cuda().contiguous()
        x = (torch.randn(Bn, 150, 3, generator=torch.Generator().manual_seed(1)) * mask[..., None]).cuda().contiguous()
        out = torch.empty_like(x)
        scratch = torch.empty(lib.pfm_epic_sample_scratch_floats(ctypes.byref(lay.desc), NI, Bn), device="cuda")
        def run():
            rc = lib.pfm_epic_sample_midpoint(ctypes.byref(lay.desc), P(blob.data_ptr()), P(ts.data_ptr()), P(dts.data_ptr()), NI,
                                              P(x.data_ptr()), P(0), P(maskd.data_ptr()), P(out.data_ptr()), Bn, P(scratch.data_ptr()), P(0))
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        key = name
        dev = ""
        if key in outs:
            dev = f"  max|dev| vs first build {(out - outs[key]).abs().max().item():.2e}"
        else:
            outs[key] = out.clone()
        print(f"{os.path.basename(path):20s} {name:20s} {e0.elapsed_time(e1)/3:8.3f} ms{dev}", flush=True)
