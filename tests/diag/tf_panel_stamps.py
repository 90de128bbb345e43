"""Diagnostic (not a test, not shipped): where does a workgroup of tf_linear_panel_kernel spend its time?  Builds a copy of the library whose
tf_kernels.hip is compiled with -DPFM_TF_DIAG (s_memtime stamps of wave 0 of every 8th workgroup of the panel launches with NO == PFM_TF_DIAG_NO),
runs cfg-4 evaluations at B = 128 and prints the phase durations of a workgroup in s_memtime ticks and as shares of its lifetime (the tick is
not calibrated here; (mean lifetime x workgroups / resident slots) against the launch's rocprofv3 duration gives ~2.2 k ticks per us).
    python tests/diag/tf_panel_stamps.py --build-only     (CPU container)
    PFM_TF_DIAG_NO=256 python tests/diag/tf_panel_stamps.py   (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from particle_fm_amd import build as B
out = os.path.join(ROOT, "tests", "diag", "libpfm_tfdiag.so")
src = os.path.join(B.CSRC, "tf_kernels.hip")
if not os.path.exists(out) or os.path.getmtime(os.path.join(B.CSRC, "tf_fwd.h")) > os.path.getmtime(out) or os.path.getmtime(src) > os.path.getmtime(out):
    B.build()
    # every translation unit that includes tf_fwd.h gets the switch (LinArgs and the inline launcher must agree across the objects)
    from concurrent.futures import ThreadPoolExecutor
    diag = [s for s in B.sources() if os.path.basename(s) in ("tf_kernels.hip", "ca_kernels.hip", "ew_kernels.hip", "mdma_kernels.hip")]
    def cc(src_):
        obj = os.path.join(ROOT, "build", os.path.basename(src_) + ".diag.o")
        subprocess.check_call([B._hipcc(), *B._flags(), "-DPFM_TF_DIAG", "-c", src_, "-o", obj])
        return obj
    with ThreadPoolExecutor(4) as ex:
        dobjs = list(ex.map(cc, diag))
    objs = [os.path.join(ROOT, "build", "obj", os.path.basename(s) + ".o") for s in B.sources() if s not in diag] + dobjs
    subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", *objs, "-o", out])
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np, torch
from particle_fm_amd import _lib
_lib.LIB_PATH = out
from particle_fm_amd import hip_ops_tf as ops
from particle_fm_amd.layout_tf import TfConfig, TfLayout
from oracle.seeded import seeded_state
Bj = 128
cfg = TfConfig(num_particles=279, global_cond_dim=5)
lay = TfLayout(cfg)
st = {k: torch.from_numpy(v) for k, v in seeded_state(dict(cfg.param_shapes()), 1).items()}
blob = lay.pack_blob(st).cuda()
gen = torch.Generator().manual_seed(0)
n = torch.randint(20, 280, (Bj,), generator=gen)
mask = (torch.arange(279)[None] < n[:, None]).float().cuda()
x = torch.randn(Bj, 279, 3, generator=gen).cuda(); cond = torch.randn(Bj, 5, generator=gen).cuda(); t = torch.rand(Bj, generator=gen).cuda()
for _ in range(4):
    ops.tf_forward(lay, blob, t, x, cond, mask)
lib = _lib.load()
lib.pfm_tf_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(512 * 8, dtype=np.uint64)
assert lib.pfm_tf_diag_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(512, 8).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
names = ["entry -> rows normalised (loads, statistics, LDS writes)", "barrier", "start values + first weights arrive", "MFMA steps of the first chunk",
         "epilogue of the first chunk", "remaining chunks"]
life = (s[:, 6] - s[:, 0]).astype(float)
print(f"NO = {os.environ.get('PFM_TF_DIAG_NO')}: {len(s)} stamped workgroups (wave 0 of every 8th), mean lifetime {life.mean():.0f} ticks")
for k, nm in enumerate(names):
    d = (s[:, k + 1] - s[:, k]).astype(float)
    print(f"  {nm:62s} mean {d.mean():8.0f} ticks = {100 * d.mean() / life.mean():5.1f} %   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f}")
