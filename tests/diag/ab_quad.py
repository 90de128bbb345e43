"""Diagnostic: the bf16 quad sampler and the bf16 one-jet sampler of two library builds on the inputs of
tests/test_hip_packed.py::test_bf16_packed_sampler_equals_unpacked_bf16_bitwise_and_meets_the_autocast_bar.
    python tests/diag/ab_quad.py libA.so libB.so     (full libraries; each runs in a child process under PFM_DIAG=1 PFM_LIB_PATH)"""
import os, subprocess, sys
import numpy as np

if os.environ.get("PFM_AB_OUT"):
    sys.path.insert(0, ".")
    import torch
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    from tests.test_layout_cpu import cfg_of
    from tests.test_hip_packed import _ragged
    g = load_golden("jetnet30")
    N, F = g.hp["num_particles"], g.hp["features"]
    out = {}
    for name, flags in (("quad16", 1 | 2 | 16), ("one16", 1 | 2), ("pair32", 1 | 16), ("one32", 1)):
        lay = EpicLayout(cfg_of(g.hp), flags=flags)
        blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
        n, mask, z, _ = _ragged(96, N, F, 0, seed=123 + 96, lo=1)
        for rep in range(2):
            out[f"{name}_{rep}"] = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=12).cpu().numpy()
    out["n"] = n.numpy()
    np.savez(os.environ["PFM_AB_OUT"], **out)
    sys.exit(0)

res = []
for i, lib in enumerate(sys.argv[1:]):
    env = dict(os.environ, PFM_LIB_PATH=os.path.abspath(lib), PFM_DIAG="1", PFM_AB_OUT=f"/tmp/ab_quad_{i}.npz")
    subprocess.run([sys.executable, __file__], check=True, env=env)
    res.append(np.load(f"/tmp/ab_quad_{i}.npz"))
a, b = res
n = a["n"]
def dj(x, y):
    return np.abs(x - y).reshape(x.shape[0], -1).max(1)
for k in ("quad16", "one16", "pair32", "one32"):
    print(f"{k}: run-to-run A {dj(a[k+'_0'], a[k+'_1']).max():.2e}  B {dj(b[k+'_0'], b[k+'_1']).max():.2e}   A vs B: jets that differ "
          f"{(dj(a[k+'_0'], b[k+'_0']) > 0).sum()} of {len(n)}, max {dj(a[k+'_0'], b[k+'_0']).max():.2e}")
for tag, r in (("A", a), ("B", b)):
    d = dj(r["quad16_0"], r["one16_0"])
    print(f"{tag}: quad16 vs one16: jets that differ {(d > 0).sum()}, max {d.max():.2e}; multiplicities of the first few: {n[d > 0][:12]}")
    d = dj(r["pair32_0"], r["one32_0"])
    print(f"{tag}: pair32 vs one32: jets that differ {(d > 0).sum()}, max {d.max():.2e}")
