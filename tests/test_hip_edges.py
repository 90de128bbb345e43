"""Edge cases of the HIP path: tiny / odd set sizes, single jet, empty batch, fully masked jet, scattered masks,
set sizes that are not a multiple of the 16-row MFMA tile, bad arguments."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, sample_midpoint
from particle_fm_amd.layout import EpicConfig, EpicLayout
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu


def _setup(N, flags=1):
    from particle_fm_amd import hip_ops
    g = load_golden("jetnet150")
    hp = dict(g.hp, num_particles=N)
    lay = EpicLayout(cfg_of(hp), flags=flags)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    vf = EpicVectorField(g.state, "flows.0.net", hp, freqs=g.freqs)
    return hip_ops, lay, blob, vf


@pytest.mark.parametrize("N", [13, 16, 17, 31, 33, 64, 97, 129, 150])
@pytest.mark.parametrize("flags", [0, 1])
def test_set_sizes_not_multiple_of_tile(N, flags):
    hip, lay, blob, vf = _setup(N, flags)
    gen = torch.Generator().manual_seed(N)
    B = 5
    n = torch.randint(1, N + 1, (B,), generator=gen)
    n[0] = N
    n[1] = 1
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, 3, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=None, mask=mask)
    v = hip.epic_forward(lay, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=1e-5, rtol=1e-4)
    xe = hip.epic_sample_midpoint(lay, blob, x.cuda(), None, mask.cuda(), ode_steps=4).cpu()
    torch.testing.assert_close(xe, sample_midpoint(vf, x, None, mask, ode_steps=4), atol=2e-5, rtol=1e-4)


def test_scattered_mask_and_single_jet():
    """The mask need not be a prefix (the kernels only use 'last valid row' to skip whole tiles)."""
    hip, lay, blob, vf = _setup(40)
    gen = torch.Generator().manual_seed(7)
    mask = (torch.rand(1, 40, 1, generator=gen) > 0.4).float()
    mask[0, 3] = 1.0
    x = torch.randn(1, 40, 3, generator=gen) * mask
    t = torch.rand(1, generator=gen)
    with torch.no_grad():
        ref = vf(t[:, None].expand(1, 40), x, cond=None, mask=mask)
    v = hip.epic_forward(lay, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=1e-5, rtol=1e-4)
    assert torch.all(v[mask.squeeze(-1) == 0] == 0)


def test_fully_masked_jet_is_nan_like_the_reference():
    """epic.py:370 divides by sum(mask): a jet without valid particles gives NaN in the reference; same here,
    and it must not disturb its neighbours."""
    hip, lay, blob, vf = _setup(32)
    gen = torch.Generator().manual_seed(3)
    mask = torch.ones(3, 32, 1)
    mask[1] = 0
    x = torch.randn(3, 32, 3, generator=gen) * mask
    t = torch.rand(3, generator=gen)
    with torch.no_grad():
        ref = vf(t[:, None].expand(3, 32), x, cond=None, mask=mask)
    v = hip.epic_forward(lay, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    assert torch.isnan(ref[1]).all() and torch.isnan(v[1]).all()
    torch.testing.assert_close(v[[0, 2]], ref[[0, 2]], atol=1e-5, rtol=1e-4)


def test_empty_batch_and_bad_arguments():
    hip, lay, blob, _ = _setup(30)
    out = hip.epic_forward(lay, blob, torch.zeros(0).cuda(), torch.zeros(0, 30, 3).cuda(), None, None)
    assert out.shape == (0, 30, 3)
    with pytest.raises(ValueError):
        hip.epic_forward(lay, blob, torch.zeros(2).cuda(), torch.zeros(2, 29, 3).cuda(), None, None)  # wrong N
    with pytest.raises(ValueError):
        hip.epic_forward(lay, blob[:-1], torch.zeros(2).cuda(), torch.zeros(2, 30, 3).cuda(), None, None)  # wrong blob
    with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
        hip.epic_forward(lay, blob, torch.zeros(2), torch.zeros(2, 30, 3), None, None)  # CPU tensors
    big = EpicLayout(EpicConfig(num_particles=200, features=3, latent=10, layers=1, frequencies=16, t_local_cat=True,
                                t_global_cat=True))
    with pytest.raises(RuntimeError, match="LDS"):
        hip.epic_forward(big, torch.zeros(big.blob_total).cuda(), torch.zeros(1).cuda(), torch.zeros(1, 200, 3).cuda(), None, None)


def test_large_batch_more_jets_than_cus():
    hip, lay, blob, vf = _setup(30)
    gen = torch.Generator().manual_seed(11)
    B = 700
    n = torch.randint(5, 31, (B,), generator=gen)
    mask = (torch.arange(30)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, 30, 3, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, 30), x, cond=None, mask=mask)
    v = hip.epic_forward(lay, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=1e-5, rtol=1e-4)
