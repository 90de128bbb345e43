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


@pytest.mark.parametrize("family", ["tf", "ca"])
def test_row_panel_linear_is_bit_identical_to_the_chunk_kernel(family):
    """Launches with >= 2 x CUs 32-row tiles run the LayerNorm-Linears on tf_linear_panel_kernel (csrc/tf_fwd.h: rows normalised once into
    LDS, barrier-free walk over the output chunks), smaller ones on tf_linear_kernel.  Same products in the same order: a jet evaluated
    inside a 72-jet batch (20 088 rows: panel kernel) must equal, bit for bit, the same jet evaluated in a batch of three (chunk kernel);
    and the big batch must match the oracle."""
    if family == "tf":
        from particle_fm_amd import hip_ops_tf as ops
        from tests.conftest import load_tf_golden
        g = load_tf_golden("lhco")
        lay, blob = _setup(g)
        fwd = ops.tf_forward
        vf = TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    else:
        from oracle.ca_ref import CrossAttentionVectorField
        from particle_fm_amd import hip_ops_ca as ops
        from particle_fm_amd.layout_ca import CaConfig, CaLayout
        from tests.conftest import load_ca_golden
        g = load_ca_golden("lhco")
        lay = CaLayout(CaConfig.from_hparams(g.hp))
        blob = lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
        fwd = ops.ca_forward
        vf = CrossAttentionVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    gen = torch.Generator().manual_seed(11)
    B, N, C, F = 72, g.hp["num_particles"], g.hp["global_cond_dim"], g.hp["features"]
    n = torch.randint(1, N + 1, (B,), generator=gen)
    n[0], n[1] = N, 1
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, F, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    cond = torch.randn(B, C, generator=gen) if C else None
    big = fwd(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    for lo in (0, 35, 69):
        sl = slice(lo, lo + 3)
        small = fwd(lay, blob, _dev(t[sl]), _dev(x[sl]), _dev(None if cond is None else cond[sl]), _dev(mask[sl])).cpu()
        assert torch.equal(big[sl], small), f"jets {lo}..{lo + 2} differ between the panel and the chunk kernel"
    # valid-rows evaluation of the same large batch (compacted rows, device-side row count, row -> jet map: the panel kernel exits past the
    # valid rows and looks the jet bias up through the map): the dense numbers at the valid particles
    lay_v = type(lay)(lay.cfg, flags=4)
    blob_v = lay_v.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    valid = fwd(lay_v, blob_v, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    keep = mask.squeeze(-1) == 1
    torch.testing.assert_close(valid[keep], big[keep], atol=1e-5, rtol=1e-4)
    sub = slice(0, 4)  # the oracle on a few jets of the big batch (CPU seconds)
    with torch.no_grad():
        ref = vf(t[sub, None].expand(4, N), x[sub], cond=None if cond is None else cond[sub], mask=mask[sub])
    torch.testing.assert_close(big[sub], ref, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("valid_rows", [False, True])
def test_two_stream_halves_of_the_midpoint_sampler_change_nothing(ops, valid_rows):
    """A midpoint call on >= 64 jets runs as two half-batches on two side streams (pfm_tf_sample_midpoint; every kernel is row- or
    jet-local): the result must be, bit for bit, what each half gives when sampled on its own (31 / 33 jets: below the split size)."""
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    from tests.conftest import load_tf_golden
    g = load_tf_golden("small")
    lay = TfLayout(TfConfig.from_hparams(g.hp), flags=4 if valid_rows else 0)  # PFM_TF_F_VALID_ROWS
    blob = lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    gen = torch.Generator().manual_seed(23)
    B, N, C, F = 65, g.hp["num_particles"], g.hp["global_cond_dim"], g.hp["features"]
    n = torch.randint(1, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen)
    cond = torch.randn(B, C, generator=gen) if C else None
    sub = lambda a, sl: None if a is None else a[sl].cuda()
    whole = ops.tf_sample_midpoint(lay, blob, z.cuda(), _dev(cond), mask.cuda(), ode_steps=4).cpu()
    h = B // 2  # the split point of the C entry
    for sl in (slice(0, h), slice(h, B)):
        part = ops.tf_sample_midpoint(lay, blob, sub(z, sl), sub(cond, sl), sub(mask, sl), ode_steps=4).cpu()
        assert torch.equal(whole[sl], part)
    # the Runge-Kutta sampler splits the same way (pfm_tf_sample_rk)
    whole_rk = ops.tf_sample_rk(lay, blob, z.cuda(), _dev(cond), mask.cuda(), ode_steps=3, solver="rk4").cpu()
    for sl in (slice(0, h), slice(h, B)):
        assert torch.equal(whole_rk[sl], ops.tf_sample_rk(lay, blob, sub(z, sl), sub(cond, sl), sub(mask, sl), ode_steps=3, solver="rk4").cpu())
    # PFM_TF_F_ONE_STREAM (callers with several calls in flight): the same call on the caller's stream alone
    lay1 = TfLayout(TfConfig.from_hparams(g.hp), flags=(4 if valid_rows else 0) | 16)
    blob1 = lay1.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    assert torch.equal(ops.tf_sample_midpoint(lay1, blob1, z.cuda(), _dev(cond), mask.cuda(), ode_steps=4).cpu(), whole)
    vf = TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
    from oracle.fm_ref import sample_midpoint
    ref = sample_midpoint(vf, z[:3], None if cond is None else cond[:3], mask[:3], ode_steps=4)
    keep = mask[:3].squeeze(-1) == 1  # padded rows: the valid-rows path never computes them
    torch.testing.assert_close(whole[:3][keep], ref[keep], atol=2e-4, rtol=1e-3)
