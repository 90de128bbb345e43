"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's golden vectors."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, sample_midpoint
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-4  # fp32 tolerance per network evaluation


def _dev(t):
    return None if t is None else t.cuda()


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from particle_fm_amd import hip_ops
    return hip_ops


def _setup(golden, flags=0):
    from particle_fm_amd.layout import EpicLayout
    lay = EpicLayout(cfg_of(golden.hp), flags=flags)
    blob = lay.pack_blob(golden.state, "flows.0.net.", freqs=golden.freqs).cuda()
    return lay, blob


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
@pytest.mark.parametrize("flags", [0, 1])
def test_forward_matches_reference_vectors(hip, golden, mk, flags):
    lay, blob = _setup(golden, flags)
    tag = f"nfe_{mk}/"
    x, t = golden.get(tag + "x"), golden.get(tag + "t")
    mask, cond = golden.get(tag + "mask"), golden.get(tag + "cond")
    v = hip.epic_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(v, golden.get(tag + "v_vec_t"), atol=ATOL, rtol=RTOL)
    if mask is not None:
        assert torch.all(v[mask.squeeze(-1) == 0] == 0)
    # scalar-t (sampling style) call: same t for every jet
    ts = t[0].expand(t.shape[0]).contiguous()
    vs = hip.epic_forward(lay, blob, _dev(ts), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(vs, golden.get(tag + "v_scalar_t"), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("steps", [3, 10, 100])
def test_midpoint_matches_reference_vectors(hip, golden, steps):
    lay, blob = _setup(golden)
    tag = f"midpoint_{steps}/"
    z, mask, cond = golden.get(tag + "z"), golden.get(tag + "mask"), golden.get(tag + "cond")
    xe = hip.epic_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=steps).cpu()
    # fp32 noise floor after 99 steps is 2.7e-6 abs on |x|~3.9
    torch.testing.assert_close(xe, golden.get(tag + "x_end"), atol=5e-5, rtol=1e-4)


def test_forward_vs_oracle_bench_shape(hip):
    """B=64 jets at the JetNet-150 shape with ragged multiplicities, against the oracle."""
    from tests.conftest import load_golden
    g = load_golden("jetnet150")
    lay, blob = _setup(g, flags=1)
    gen = torch.Generator().manual_seed(1)
    B, N = 64, 150
    n = torch.randint(30, 151, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, 3, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=None, mask=mask)
    v = hip.epic_forward(lay, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=ATOL, rtol=RTOL)
    # permutation equivariance of the valid particles (SURVEY.md §4 property)
    perm = torch.stack([torch.cat([torch.randperm(int(k), generator=gen), torch.arange(int(k), N)]) for k in n])
    xp = torch.gather(x, 1, perm[..., None].expand(-1, -1, 3))
    vp = hip.epic_forward(lay, blob, t.cuda(), xp.cuda(), None, mask.cuda()).cpu()
    torch.testing.assert_close(vp, torch.gather(v, 1, perm[..., None].expand(-1, -1, 3)), atol=2e-6, rtol=1e-5)


def test_time_term_table_is_a_pure_optimisation(hip, golden):
    """The sampler tabulates the time columns of the per-jet Linears once per call (all jets share the evaluation times) and
    skips those weight rows in the kernel: same trajectory as the kernel that fetches them (sum order differs: ~1e-7)."""
    lay, blob = _setup(golden)
    tag = "midpoint_10/"
    z, mask, cond = golden.get(tag + "z"), golden.get(tag + "mask"), golden.get(tag + "cond")
    a = hip.epic_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10, time_table=True).cpu()
    b = hip.epic_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10, time_table=False).cpu()
    torch.testing.assert_close(a, b, atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(a, golden.get(tag + "x_end"), atol=5e-5, rtol=1e-4)


def test_jet_launch_order_is_longest_first(hip, golden):
    """The sampler's scratch ends with the workgroup list of the call, [n_workgroups, (jet, partner or -1) ...]: without
    PFM_F_PACK_JETS one jet per workgroup -- a permutation, descending multiplicity, ties by index."""
    import ctypes
    from particle_fm_amd import _lib
    if golden.get("midpoint_10/mask") is None:
        pytest.skip("no mask in this fixture")
    lay, blob = _setup(golden)
    gen = torch.Generator().manual_seed(21)
    B, N, F = 300, golden.hp["num_particles"], golden.hp["features"]   # more jets than CUs
    n = torch.randint(1, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen)
    cond = None if golden.get("midpoint_10/cond") is None else torch.randn(B, golden.get("midpoint_10/cond").shape[1], generator=gen)
    out = hip.epic_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=4)
    torch.cuda.synchronize()
    scratch = [v for k, v in lay.__dict__["_sample_scratch"].items() if k[0] == 4 and k[3] == B][0]
    total = _lib.load().pfm_epic_sample_scratch_floats(ctypes.byref(lay.desc), 3, B)
    ints = scratch.view(torch.int32)[total - ((2 * B + 1 + 63) // 64) * 64:].cpu().long()
    assert int(ints[0]) == B
    wl = ints[1:1 + 2 * B].reshape(B, 2)
    assert torch.all(wl[:, 1] == -1)  # no pairing unless asked for
    order = wl[:, 0]
    assert sorted(order.tolist()) == list(range(B))
    key = n[order]
    assert torch.all(key[:-1] >= key[1:])
    same = key[:-1] == key[1:]
    assert torch.all(order[:-1][same] < order[1:][same])
    # and the order is scheduling only: every jet equals the jet sampled on its own
    for jj in (int(order[0]), int(order[-1]), 7):
        one = hip.epic_sample_midpoint(lay, blob, _dev(z[jj:jj + 1]), _dev(None if cond is None else cond[jj:jj + 1]), _dev(mask[jj:jj + 1]),
                                       ode_steps=4)
        torch.testing.assert_close(out[jj].cpu(), one[0].cpu(), atol=0, rtol=0)
