"""Parity of the wide EPiC HIP path (hidden 300, JetClass configuration) with the reference's recorded vectors."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu


def _dev(t):
    return None if t is None else t.cuda()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from particle_fm_amd import hip_ops_wide
    return hip_ops_wide


def _setup(g):
    from particle_fm_amd.layout_wide import EpicWideLayout
    lay = EpicWideLayout(cfg_of(g.hp))
    return lay, lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_forward_matches_reference_vectors(ops, wide_golden, mk):
    g = wide_golden
    lay, blob = _setup(g)
    tag = f"nfe_{mk}/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ops.ew_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)
    if mask is not None:
        assert torch.all(v[mask.squeeze(-1) == 0] == 0)
    vs = ops.ew_forward(lay, blob, _dev(t[0]), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(vs, g.get(tag + "v_scalar_t"), atol=2e-5, rtol=2e-4)


@pytest.mark.parametrize("steps", [3, 10])
def test_midpoint_matches_reference_vectors(ops, wide_golden, steps):
    g = wide_golden
    lay, blob = _setup(g)
    tag = f"midpoint_{steps}/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    xe = ops.ew_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=steps).cpu()
    torch.testing.assert_close(xe, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)


def test_narrow_config_agrees_with_the_jet_resident_kernel(ops, golden):
    """hidden 128 through the wide path = the jet-resident kernel's answer = the reference's."""
    lay, blob = _setup(golden)
    tag = "nfe_f32/"
    x, t, mask, cond = (golden.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ops.ew_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(v, golden.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)


def test_scattered_masks_and_empty_jet(ops):
    from tests.conftest import load_wide_golden
    g = load_wide_golden("small")
    lay, blob = _setup(g)
    gen = torch.Generator().manual_seed(5)
    B, N, Fe, C = 7, g.hp["num_particles"], g.hp["features"], g.hp["global_cond_dim"]
    mask = (torch.rand(B, N, 1, generator=gen) < 0.5).float()
    mask[:, 0] = 1.0
    mask[3] = 0.0  # an empty jet: the reference's mean is 0/0 -> NaN rows, other jets unaffected
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, C, generator=gen)
    t = torch.rand(B, generator=gen)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=cond, mask=mask)
    v = ops.ew_forward(lay, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    assert torch.isnan(ref[3]).all() and torch.isnan(v[3]).all()
    keep = [i for i in range(B) if i != 3]
    torch.testing.assert_close(v[keep], ref[keep], atol=2e-5, rtol=2e-4)


def test_two_stream_sampler_equals_separate_halves(ops):
    """From 64 jets on the midpoint sampler runs two half-batches on two streams (interleaved launches, side stream joined at
    the end): the result must be what the two halves give on their own."""
    from tests.conftest import load_wide_golden
    g = load_wide_golden("small")
    lay, blob = _setup(g)
    gen = torch.Generator().manual_seed(8)
    B, N, F, C = 70, g.hp["num_particles"], g.hp["features"], g.hp["global_cond_dim"]
    n = torch.randint(2, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen) * mask
    cond = torch.randn(B, C, generator=gen)
    full = ops.ew_sample_midpoint(lay, blob, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=6).cpu()
    a = ops.ew_sample_midpoint(lay, blob, z[:35].cuda(), cond[:35].cuda(), mask[:35].cuda(), ode_steps=6).cpu()
    b = ops.ew_sample_midpoint(lay, blob, z[35:].cuda(), cond[35:].cuda(), mask[35:].cuda(), ode_steps=6).cpu()
    torch.testing.assert_close(full, torch.cat([a, b]), atol=1e-6, rtol=1e-6)
    # and the caller's stream really waits for the side stream: a dependent op right after sees the finished result
    out = ops.ew_sample_midpoint(lay, blob, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=6)
    s = out.sum()
    torch.testing.assert_close(s.cpu(), full.sum(), atol=1e-3, rtol=1e-5)


def test_results_do_not_depend_on_the_batch_size(ops):
    """The Linear launches pick their tiles (and, on the transformer paths, their kernels) by the number of rows: a jet evaluated inside a
    160-jet batch (640 row tiles of 32) must equal, bit for bit, the same jet in a batch of three, on dense rows and with a mask (compacted
    rows); and the large batch must match the oracle.  (A one-launch fusion of a layer's two local Linears was measured with this test and
    dropped: DESIGN 7.)"""
    from tests.conftest import load_wide_golden
    g = load_wide_golden("jetclass")
    lay, blob = _setup(g)
    gen = torch.Generator().manual_seed(19)
    B, N, F = 160, g.hp["num_particles"], g.hp["features"]
    C = g.hp["global_cond_dim"]
    n = torch.randint(1, N + 1, (B,), generator=gen)
    n[0], n[1] = N, 1
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, F, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    cond = torch.randn(B, C, generator=gen) if C else None
    for mk in (None, mask):
        big = ops.ew_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mk)).cpu()
        for lo in (0, 77, 157):
            sl = slice(lo, lo + 3)
            small = ops.ew_forward(lay, blob, _dev(t[sl]), _dev(x[sl]), _dev(None if cond is None else cond[sl]), _dev(None if mk is None else mk[sl])).cpu()
            assert torch.equal(big[sl], small), f"jets {lo}..{lo + 2} differ between the large and the small batch"
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    with torch.no_grad():
        ref = vf(t[:4, None].expand(4, N), x[:4], cond=None if cond is None else cond[:4], mask=mask[:4])
    torch.testing.assert_close(big[:4], ref, atol=2e-5, rtol=2e-4)
