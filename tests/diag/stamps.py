"""Diagnostic (not a test, not shipped): build a -DPFM_DIAG copy of the library into gpurun_out/,
run ONE forward launch and print where workgroup 0 spent its cycles (s_memtime, 100 MHz ticks)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, ".")
import torch
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd import build as B

# The stamps live in epic_kernels.hip only, which is self-contained: one translation unit, no -fgpu-rdc.  Built here (CPU container,
# `python tests/diag/stamps.py --build-only`) so that the GPU box does not spend its minutes compiling; rebuilt when stale.
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PFM_DIAG_LIB", "libpfm_diag.so"))  # PFM_DIAG_LIB + PFM_DEFS: variants side by side
src = os.path.join(B.CSRC, "epic_kernels.hip")
deps = [src] + [os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".h")]
if not os.path.exists(out) or any(os.path.getmtime(d_) > os.path.getmtime(out) for d_ in deps):
    cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPFM_DIAG", *os.environ.get("PFM_DEFS", "").split(),
           "-Iinclude", "-Iparticle_fm_amd/csrc", src, "-o", out]
    subprocess.check_call(cmd)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(out)
g = load_golden("jetnet150")
NP = int(os.environ.get("PFM_N", "150"))
hp = dict(g.hp); hp["num_particles"] = NP
lay = EpicLayout(cfg_of(hp), flags=int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else (1 if os.environ.get("PFM_MASKN") else 0))
blob = lay.pack_blob(g.state, "flows.0.net.").cuda()
Bn = 256
gen = torch.Generator().manual_seed(0)
x = torch.randn(Bn, NP, 3, generator=gen).cuda(); t = torch.rand(Bn, generator=gen).cuda(); v = torch.empty_like(x)
P = ctypes.c_void_p
MODE = os.environ.get("PFM_MODE", "sample")  # "forward": one evaluation launch
from particle_fm_amd.hip_ops import midpoint_grid
ts, dts = midpoint_grid(6)
ts, dts = ts.cuda(), dts.cuda()
lib.pfm_epic_sample_scratch_floats.restype = ctypes.c_int64
scratch = torch.empty(lib.pfm_epic_sample_scratch_floats(ctypes.byref(lay.desc), 5, Bn), device="cuda")  # time-term table (the bench's variant)
SCR = P(0) if os.environ.get("PFM_NO_TB") else P(scratch.data_ptr())
MASKN = int(os.environ.get("PFM_MASKN", "0"))  # > 0: every jet has MASKN valid particles (masked tail skipped; short jets pair up)
maskt = None
if MASKN:
    maskt = (torch.arange(NP)[None] < torch.full((Bn, 1), MASKN)).float().cuda().contiguous()
MSK = P(maskt.data_ptr()) if MASKN else P(0)
for it in range(3):
    if MODE == "forward":
        rc = lib.pfm_epic_forward(ctypes.byref(lay.desc), P(blob.data_ptr()), P(t.data_ptr()), P(x.data_ptr()), P(0), P(0), P(v.data_ptr()), Bn, P(0), P(0))
    else:  # stamps of the LAST of 10 evaluations inside the persistent sampler (warm scalar cache, steady state)
        rc = lib.pfm_epic_sample_midpoint(ctypes.byref(lay.desc), P(blob.data_ptr()), P(ts.data_ptr()), P(dts.data_ptr()), 5,
                                          P(x.data_ptr()), P(0), MSK, P(v.data_ptr()), Bn, SCR, P(0))
    assert rc == 0, rc
    buf = (ctypes.c_ulonglong * 512)(); n = ctypes.c_int(0)
    lib.pfm_diag_read_stamps(buf, ctypes.byref(n))
names = {0: "start", 1: "body", 2: "stem-bias done", 3: "fc_l1 done", 4: "fc_l2+pool done", 10: "layer: global start", 11: "layer: global done",
         12: "layer: bias done", 41: "pj: S2 done", 42: "pj: S3 done", 43: "pj: S4 done", 44: "pj: S5 done", 13: "layer: phase1 done", 20: "layers done", 30: "head done"}
prev = None; tot = {}
for i in range(n.value):
    sid, tk = buf[2 * i] >> 48, buf[2 * i + 1]
    if prev is not None:
        key = f"{names.get(prev[0], prev[0])} -> {names.get(sid, sid)}"
        tot.setdefault(key, []).append(tk - prev[1])
    prev = (sid, tk)
total = buf[2 * (n.value - 1) + 1] - buf[1]
rt = (buf[2 * (n.value - 1)] & 0xFFFFFFFFFFFF) - (buf[0] & 0xFFFFFFFFFFFF)
print(f"s_memrealtime ticks {rt} (100 MHz => {rt/100:.1f} us); s_memtime/s_memrealtime*100MHz = {total/rt*100:.0f} MHz")
print(f"total ticks {total} (100 MHz => {total/100:.1f} us)")
for k, v_ in tot.items():
    print(f"{k:50s} n={len(v_):2d} mean {sum(v_)/len(v_):8.1f} ticks  sum {sum(v_):7d}  ({100*sum(v_)/total:.1f}%)")

marks = [buf[384 + i] for i in range(8)]
deltas = [marks[i + 1] - marks[i] for i in range(6)]
if all(0 <= d_ < 1 << 24 for d_ in deltas):  # the generic chain's PFM_MARKs; the lean chain (epic_fast.h) sets none
    print("per-jet marks (last layer), deltas:", deltas)
print("NOTE: the -DPFM_DIAG build runs ~25 % slower than the shipped one (its stamp stores cut the scheduling regions): use these numbers "
      "for proportions; absolute per-piece costs come from ablation builds (tests/diag/ab_time.py)")
