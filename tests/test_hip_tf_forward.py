"""Parity of the HIP Full-Transformer path (through the C ABI) with the reference's recorded vectors and the oracle."""
import pytest
import torch

from oracle.tf_ref import TransformerVectorField

pytestmark = pytest.mark.gpu

ATOL, RTOL = 2e-5, 2e-4  # fp32 tolerance per network evaluation (|v| ~ 1)


def _dev(t):
    return None if t is None else t.cuda()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from particle_fm_amd import hip_ops_tf
    return hip_ops_tf


def _setup(g):
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    lay = TfLayout(TfConfig.from_hparams(g.hp))
    return lay, lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()


@pytest.mark.parametrize("mk", ["f32", "int64", "ones"])
def test_forward_matches_reference_vectors(ops, tf_golden, mk):
    g = tf_golden
    lay, blob = _setup(g)
    tag = f"nfe_{mk}/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ops.tf_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=ATOL, rtol=RTOL)
    vs = ops.tf_forward(lay, blob, _dev(t[0]), _dev(x), _dev(cond), _dev(mask)).cpu()  # 0-dim t of sampling
    torch.testing.assert_close(vs, g.get(tag + "v_scalar_t"), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("steps", [3, 10, 100])
def test_midpoint_matches_reference_vectors(ops, tf_golden, steps):
    g = tf_golden
    tag = f"midpoint_{steps}/"
    if g.get(tag + "z") is None:
        pytest.skip("not recorded at this size")
    lay, blob = _setup(g)
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    xe = ops.tf_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=steps).cpu()
    torch.testing.assert_close(xe, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)


def test_forward_vs_oracle_ragged_batch(ops):
    """B = 19 jets (rows not a multiple of the 64-row tile) with scattered key masks, against the oracle."""
    from tests.conftest import load_tf_golden
    g = load_tf_golden("small")
    lay, blob = _setup(g)
    gen = torch.Generator().manual_seed(3)
    B, N, C = 19, g.hp["num_particles"], g.hp["global_cond_dim"]
    mask = (torch.rand(B, N, 1, generator=gen) < 0.6).float()
    mask[:, 0] = 1.0
    x = torch.randn(B, N, 3, generator=gen)
    cond = torch.randn(B, C, generator=gen)
    t = torch.rand(B, generator=gen)
    vf = TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=cond, mask=mask)
    v = ops.tf_forward(lay, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=ATOL, rtol=RTOL)
    # valid particles are permutation equivariant; padded keys never influence anyone
    perm = torch.stack([torch.randperm(N, generator=gen) for _ in range(B)])
    gat = lambda a: torch.gather(a, 1, perm[..., None].expand(-1, -1, a.shape[-1]))
    vp = ops.tf_forward(lay, blob, t.cuda(), gat(x).cuda(), cond.cuda(), gat(mask).cuda()).cpu()
    torch.testing.assert_close(vp, gat(v), atol=5e-6, rtol=1e-4)
    x2 = x + (1 - mask) * torch.randn(B, N, 3, generator=gen)
    v2 = ops.tf_forward(lay, blob, t.cuda(), x2.cuda(), cond.cuda(), mask.cuda()).cpu()
    keep = mask.squeeze(-1) == 1
    torch.testing.assert_close(v2[keep], v[keep], atol=0, rtol=0)
