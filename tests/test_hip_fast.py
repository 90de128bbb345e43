"""The midpoint sampler's lean evaluation (csrc/epic_fast.h: unconditioned jets, T = 32, F <= 4): same results as the generic
kernel up to fp32 re-association of the tabulated stem terms, and the reference's vectors / the oracle within the usual bars.

Reference: particle_fm/models/components/epic.py:304-391, :85-203; flow_matching_module.py:245-259 (midpoint decode)."""
import ctypes

import pytest
import torch

from oracle.fm_ref import EpicVectorField, sample_midpoint
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu

SKIP_TAIL, BF16, PACK, GENERIC = 1, 2, 16, 32


def _ragged(B, N, F, seed, lo=1, hi=None):
    gen = torch.Generator().manual_seed(seed)
    n = torch.randint(lo, (hi or N) + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen)
    return n, mask, z


def _layouts(name, extra=0):
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden(name)
    fast = EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | extra)
    gen = EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | GENERIC | extra)
    blob_f = fast.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    blob_g = gen.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    return g, fast, blob_f, gen, blob_g


def test_path_selection():
    """Which descriptors take the lean evaluation: the headline configuration does, conditioned / packed / split-fp16 ones do not."""
    from particle_fm_amd import _lib
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    lib = _lib.load()
    q = lambda lay: lib.pfm_epic_sample_is_fast(ctypes.byref(lay.desc))
    g = load_golden("jetnet150")
    assert q(EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL)) == 1
    assert q(EpicLayout(cfg_of(g.hp), flags=0)) == 1
    assert q(EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | BF16)) == 1
    assert q(EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | GENERIC)) == 0
    assert q(EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | PACK)) == 1  # two jets per workgroup run on the lean evaluation too
    assert q(EpicLayout(cfg_of(g.hp), flags=SKIP_TAIL | 4)) == 0  # split fp16
    assert q(EpicLayout(cfg_of(load_golden("jetnet30").hp), flags=SKIP_TAIL)) == 1
    assert q(EpicLayout(cfg_of(load_golden("cond_gl").hp), flags=SKIP_TAIL)) == 1  # conditioned (2 + 2): the jet's terms from the cond table
    assert q(EpicLayout(cfg_of(load_golden("cond_gl").hp), flags=SKIP_TAIL | PACK)) == 0  # ... one jet per workgroup only
    assert q(EpicLayout(cfg_of(load_golden("cond_jetclass").hp), flags=SKIP_TAIL)) == 0  # 13 features: fc_l1 is not one MFMA


@pytest.mark.parametrize("name,B,steps", [("jetnet150", 40, 6), ("jetnet30", 48, 11), ("jetnet150", 7, 100)])
def test_fast_sampler_matches_generic_and_oracle(name, B, steps):
    from particle_fm_amd import hip_ops
    g, fast, blob_f, gen, blob_g = _layouts(name)
    N, F = g.hp["num_particles"], g.hp["features"]
    n, mask, z = _ragged(B, N, F, seed=7 + B)
    n[0], n[1] = N, 1  # a full jet and a one-particle jet
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    a = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
    b = hip_ops.epic_sample_midpoint(gen, blob_g, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
    # same arithmetic per layer; the tabulated stem terms and the MFMA fc_l1 re-associate fp32 sums (~1e-7 per evaluation)
    torch.testing.assert_close(a, b, atol=5e-6, rtol=1e-5)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_midpoint(vf, z, None, mask, ode_steps=steps)
    torch.testing.assert_close(a, ref, atol=5e-5 if steps > 20 else 2e-5, rtol=1e-4)
    assert torch.all(a[mask.squeeze(-1) == 0] == 0)  # masked rows exactly 0
    assert torch.equal(a, hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu())


def test_fast_sampler_against_reference_vectors():
    """The fixtures' midpoint end states (reference vector field + restated integrator) at 3 / 10 / 100 steps."""
    from particle_fm_amd import hip_ops
    for name in ("jetnet150", "jetnet30"):
        g, fast, blob_f, _, _ = _layouts(name)
        for steps in (3, 10, 100):
            tag = f"midpoint_{steps}/"
            z, mask, want = g.get(tag + "z"), g.get(tag + "mask"), g.get(tag + "x_end")
            assert z is not None and want is not None
            out = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, None if mask is None else mask.float().cuda(),
                                               ode_steps=steps).cpu()
            torch.testing.assert_close(out, want, atol=5e-5, rtol=1e-4)


def test_fast_sampler_without_mask_and_without_tail_skipping():
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("jetnet30")
    N, F = g.hp["num_particles"], g.hp["features"]
    lay = EpicLayout(cfg_of(g.hp), flags=0)  # every row computed
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    _, mask, z = _ragged(16, N, F, seed=3)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    out = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, None, ode_steps=5).cpu()
    torch.testing.assert_close(out, sample_midpoint(vf, z, None, None, ode_steps=5), atol=2e-5, rtol=1e-4)
    out = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=5).cpu()
    torch.testing.assert_close(out, sample_midpoint(vf, z, None, mask, ode_steps=5), atol=2e-5, rtol=1e-4)


def test_fast_sampler_all_masked_jet_is_nan_like_the_reference():
    """A jet without a valid particle divides 0 / 0 in the pooled mean (epic.py:370); its neighbours are untouched."""
    from particle_fm_amd import hip_ops
    g, fast, blob_f, gen, blob_g = _layouts("jetnet30")
    N, F = g.hp["num_particles"], g.hp["features"]
    _, mask, z = _ragged(6, N, F, seed=11)
    mask[2] = 0
    a = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=4).cpu()
    b = hip_ops.epic_sample_midpoint(gen, blob_g, z.cuda(), None, mask.cuda(), ode_steps=4).cpu()
    keep = [0, 1, 3, 4, 5]
    torch.testing.assert_close(a[keep], b[keep], atol=5e-6, rtol=1e-5)
    assert torch.equal(torch.isnan(a[2]), torch.isnan(b[2]))


def test_fast_bf16_sampler_tracks_the_generic_bf16_kernel():
    from particle_fm_amd import hip_ops
    g, fast, blob_f, gen, blob_g = _layouts("jetnet150", extra=BF16)
    N, F = g.hp["num_particles"], g.hp["features"]
    _, mask, z = _ragged(24, N, F, seed=19, lo=4)
    a = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=8).cpu()
    b = hip_ops.epic_sample_midpoint(gen, blob_g, z.cuda(), None, mask.cuda(), ode_steps=8).cpu()
    # both round the same operands to bf16; inputs that differ by 1e-7 may round to neighbouring bf16 values
    torch.testing.assert_close(a, b, atol=3e-3, rtol=3e-3)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_midpoint(vf, z, None, mask, ode_steps=8)
    assert (a - ref).abs().max() < 1.5 * (b - ref).abs().max() + 1e-3


def test_lds_boundary_of_the_fast_path():
    """The chain's table rows live in 1152 bytes of LDS behind the activation tile: at F = 3 that fits up to 150 particles;
    a 151-particle set (which the generic kernel still holds) stays on the generic kernel."""
    from particle_fm_amd import _lib
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    lib = _lib.load()
    g = load_golden("jetnet150")
    for n, want in ((150, 1), (151, 0), (128, 1)):
        hp = dict(g.hp); hp["num_particles"] = n
        lay = EpicLayout(cfg_of(hp), flags=SKIP_TAIL)
        assert lib.pfm_epic_sample_is_fast(ctypes.byref(lay.desc)) == want, n


def test_fast_sampler_properties_at_the_bench_size():
    """Size-independent properties at the bench's full size (256 jets, N = 150, 100 midpoint steps, multiplicities U{30..150}), where
    the oracle would take minutes: a jet's result does not depend on its batch (bitwise), the field is permutation equivariant
    (epic.py pools with a masked mean / sum), masked rows are exactly 0, everything is finite."""
    from particle_fm_amd import hip_ops
    g, fast, blob_f, _, _ = _layouts("jetnet150")
    N, F, B = 150, 3, 256
    gen = torch.Generator().manual_seed(2024)
    n = torch.randint(30, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen)
    out = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=100).cpu()
    assert torch.isfinite(out).all()
    assert torch.all(out[mask.squeeze(-1) == 0] == 0)
    # batch independence: 40 jets picked out of the batch, in another order
    pick = torch.randperm(B, generator=gen)[:40]
    sub = hip_ops.epic_sample_midpoint(fast, blob_f, z[pick].cuda(), None, mask[pick].cuda(), ode_steps=100).cpu()
    assert torch.equal(sub, out[pick])
    # permutation equivariance over the valid particles (the sums re-associate: fp32 noise amplified over 198 evaluations)
    zp, perm = z.clone(), []
    for b in range(B):
        p = torch.randperm(int(n[b]), generator=gen)
        perm.append(p)
        zp[b, : int(n[b])] = z[b, p]
    outp = hip_ops.epic_sample_midpoint(fast, blob_f, zp.cuda(), None, mask.cuda(), ode_steps=100).cpu()
    for b in range(0, B, 7):
        torch.testing.assert_close(outp[b, : int(n[b])], out[b, perm[b]], atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("solver,steps", [("euler", 9), ("rk4", 6), ("midpoint", 12)])
def test_fast_rk_sampler_matches_generic_and_oracle(solver, steps):
    """pfm_epic_sample_rk_sized with the full scratch runs the lean evaluation for unconditioned jets: against the generic Runge-Kutta
    kernel (PFM_F_GENERIC_SAMPLER) and the oracle's restated torchdyn steppers (oracle/fm_ref.py::sample_fixed_step)."""
    from oracle.fm_ref import sample_fixed_step
    from particle_fm_amd import hip_ops
    g, fast, blob_f, gen, blob_g = _layouts("jetnet150")
    N, F = g.hp["num_particles"], g.hp["features"]
    n, mask, z = _ragged(20, N, F, seed=5 + steps)
    n[0], n[1] = N, 1
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    a = hip_ops.epic_sample_rk(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=steps, solver=solver).cpu()
    b = hip_ops.epic_sample_rk(gen, blob_g, z.cuda(), None, mask.cuda(), ode_steps=steps, solver=solver).cpu()
    torch.testing.assert_close(a, b, atol=5e-6, rtol=1e-5)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_fixed_step(vf, z, None, mask, ode_steps=steps, solver=solver)
    torch.testing.assert_close(a, ref, atol=2e-5, rtol=1e-4)
    assert torch.all(a[mask.squeeze(-1) == 0] == 0)
    assert torch.equal(a, hip_ops.epic_sample_rk(fast, blob_f, z.cuda(), None, mask.cuda(), ode_steps=steps, solver=solver).cpu())


@pytest.mark.parametrize("B,steps", [(24, 6), (5, 100)])
def test_fast_sampler_conditioned_jets(B, steps):
    """global + local conditioning (fm_tops*_cond.yaml: 2 + 2): the conditioning columns of every per-jet Linear come from the
    per-jet cond table (epic_cond_table_kernel); against the generic kernel, the oracle and the reference's midpoint vectors."""
    from particle_fm_amd import hip_ops
    g, fast, blob_f, gen, blob_g = _layouts("cond_gl")
    N, F, C = g.hp["num_particles"], g.hp["features"], g.hp["global_cond_dim"]
    n, mask, z = _ragged(B, N, F, seed=3 + B)
    n[0], n[1] = N, 1
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    cond = torch.randn(B, C, generator=torch.Generator().manual_seed(B))
    a = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=steps).cpu()
    b = hip_ops.epic_sample_midpoint(gen, blob_g, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=steps).cpu()
    torch.testing.assert_close(a, b, atol=5e-6, rtol=1e-5)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_midpoint(vf, z, cond, mask, ode_steps=steps)
    torch.testing.assert_close(a, ref, atol=5e-5 if steps > 20 else 2e-5, rtol=1e-4)
    assert torch.all(a[mask.squeeze(-1) == 0] == 0)
    # a jet's result depends on its own conditioning only
    cond2 = cond.clone(); cond2[3:] += 1.0
    a2 = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), cond2.cuda(), mask.cuda(), ode_steps=steps).cpu()
    assert torch.equal(a2[:3], a[:3]) and not torch.equal(a2[3:], a[3:])
    for st in (3, 10):
        tag = f"midpoint_{st}/"
        out = hip_ops.epic_sample_midpoint(fast, blob_f, g.get(tag + "z").cuda(), g.get(tag + "cond").cuda(),
                                           None if g.get(tag + "mask") is None else g.get(tag + "mask").float().cuda(), ode_steps=st).cpu()
        torch.testing.assert_close(out, g.get(tag + "x_end"), atol=5e-5, rtol=1e-4)
