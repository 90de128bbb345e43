"""Diagnostic: the cfg-2 sampler (EPiC JetNet-30, 1024 jets, n ~ U{10..30}, 100-step midpoint) under several FULL library builds, bf16 quad
kernel and fp32 pair kernel.    python tests/diag/ab_cfg2.py libA.so libB.so ...   (child process per library: PFM_DIAG=1 PFM_LIB_PATH)"""
import os, subprocess, sys

if os.environ.get("PFM_AB_CHILD"):
    sys.path.insert(0, ".")
    import torch
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    from tests.test_layout_cpu import cfg_of
    g = load_golden("jetnet30")
    N, F, B = g.hp["num_particles"], g.hp["features"], 1024
    gen = torch.Generator().manual_seed(7)
    n = torch.randint(10, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float()
    z = (torch.randn(B, N, F, generator=gen) * mask[..., None]).cuda()
    mask = mask.cuda()
    for name, flags in (("bf16 quad", 1 | 2 | 16), ("fp32 pairs", 1 | 16), ("bf16 one jet", 1 | 2)):
        lay = EpicLayout(cfg_of(g.hp), flags=flags)
        blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
        out = hip_ops.epic_sample_midpoint(lay, blob, z, None, mask, ode_steps=100)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = hip_ops.epic_sample_midpoint(lay, blob, z, None, mask, ode_steps=100)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{os.path.basename(os.environ['PFM_LIB_PATH']):28s} {name:14s} {ms:8.3f} ms  {B / ms:7.1f} k jets/s  checksum {float(out.double().abs().sum()):.6f}", flush=True)
    sys.exit(0)

for lib in sys.argv[1:]:
    env = dict(os.environ, PFM_LIB_PATH=os.path.abspath(lib), PFM_DIAG="1", PFM_AB_CHILD="1")
    subprocess.run([sys.executable, __file__], check=True, env=env, stderr=subprocess.DEVNULL)
