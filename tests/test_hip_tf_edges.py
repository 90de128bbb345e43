"""Edge cases of the transformer and wide-EPiC HIP paths: empty batches, odd set sizes, missing masks, bad arguments."""
import copy

import pytest
import torch

from oracle.fm_ref import EpicVectorField
from oracle.seeded import seeded_state
from oracle.tf_ref import TransformerVectorField

pytestmark = pytest.mark.gpu


def _tf(N, C=2, D=128, layers=1, heads=8):
    from particle_fm_amd.layout_tf import TfConfig, TfLayout, default_freqs
    cfg = TfConfig(num_particles=N, model_dim=D, num_layers=layers, num_heads=heads, hidden=2 * D, ctxt_hidden=2 * D,
                   global_cond_dim=C)
    lay = TfLayout(cfg)
    st = {k: torch.from_numpy(v) for k, v in seeded_state(dict(cfg.param_shapes()), 5).items()}
    hp = dict(frequencies=16, add_time_to_input=True, net_config=dict(te_config=dict(
        num_layers=layers, mha_config=dict(num_heads=heads, do_layer_norm=True), dense_config=dict(nrm="layer"))))
    vf = TransformerVectorField(st, "", hp, freqs=default_freqs(32))
    return lay, lay.pack_blob(st).cuda(), vf


@pytest.mark.parametrize("N", [2, 17, 33, 200])
def test_transformer_odd_set_sizes_and_no_mask(N):
    from particle_fm_amd import hip_ops_tf as ops
    lay, blob, vf = _tf(N)
    gen = torch.Generator().manual_seed(N)
    B = 3
    x, cond, t = torch.randn(B, N, 3, generator=gen), torch.randn(B, 2, generator=gen), torch.rand(B, generator=gen)
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=cond, mask=torch.ones(B, N, 1))
    v = ops.tf_forward(lay, blob, t.cuda(), x.cuda(), cond.cuda(), None).cpu()  # mask None = all valid
    torch.testing.assert_close(v, ref, atol=2e-5, rtol=2e-4)


def test_transformer_fully_masked_jet_is_nan_like_sdpa():
    from particle_fm_amd import hip_ops_tf as ops
    lay, blob, vf = _tf(20)
    gen = torch.Generator().manual_seed(1)
    B, N = 3, 20
    x, cond, t = torch.randn(B, N, 3, generator=gen), torch.randn(B, 2, generator=gen), torch.rand(B, generator=gen)
    mask = torch.ones(B, N, 1)
    mask[1] = 0
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=cond, mask=mask)
    v = ops.tf_forward(lay, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    assert torch.isnan(ref[1]).all() and torch.isnan(v[1]).all()
    torch.testing.assert_close(v[[0, 2]], ref[[0, 2]], atol=2e-5, rtol=2e-4)


def test_empty_batch_and_bad_arguments():
    from particle_fm_amd import hip_ops_tf as ops, hip_ops_wide as opw
    from particle_fm_amd.layout import EpicConfig
    from particle_fm_amd.layout_wide import EpicWideLayout
    lay, blob, _ = _tf(12)
    out = ops.tf_forward(lay, blob, torch.zeros(0).cuda(), torch.zeros(0, 12, 3).cuda(), torch.zeros(0, 2).cuda(), None)
    assert out.shape == (0, 12, 3)
    with pytest.raises(ValueError):
        ops.tf_forward(lay, blob, torch.zeros(2).cuda(), torch.zeros(2, 12, 3).cuda(), None, None)  # cond missing
    with pytest.raises(ValueError):
        ops.tf_forward(lay, blob, torch.zeros(2).cuda(), torch.zeros(2, 13, 3).cuda(), torch.zeros(2, 2).cuda(), None)
    with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
        ops.tf_forward(lay, blob, torch.zeros(2), torch.zeros(2, 12, 3), torch.zeros(2, 2), None)
    cfg = EpicConfig(num_particles=9, features=4, hidden_dim=200, latent=7, layers=1, frequencies=16, t_local_cat=True,
                     t_global_cat=True, global_cond_dim=0, local_cond_dim=0)
    wl = EpicWideLayout(cfg)
    shapes = {}
    for name, i, o in cfg.linear_shapes():
        shapes[name + ".bias"], shapes[name + ".weight_g"], shapes[name + ".weight_v"] = (o,), (o, 1), (o, i)
    st = {k: torch.from_numpy(v) for k, v in seeded_state(shapes, 3).items()}
    wblob = wl.pack_blob(st).cuda()
    assert opw.ew_forward(wl, wblob, torch.zeros(0).cuda(), torch.zeros(0, 9, 4).cuda(), None, None).shape == (0, 9, 4)
    # odd everything (N = 9, F = 4, H = 200 -> padded to 256, L = 7), no mask, no cond, shared time
    gen = torch.Generator().manual_seed(2)
    x, t = torch.randn(5, 9, 4, generator=gen), torch.rand(1, generator=gen)
    hp = dict(frequencies=16, layers=1, t_local_cat=True, t_global_cat=True, global_cond_dim=0, local_cond_dim=0, sum_scale=1e-2)
    vf = EpicVectorField(st, "", hp, freqs=wl.default_freqs())
    with torch.no_grad():
        ref = vf(t[0], x, cond=None, mask=None)
    v = opw.ew_forward(wl, wblob, t.cuda(), x.cuda(), None, None).cpu()
    torch.testing.assert_close(v, ref, atol=2e-5, rtol=2e-4)


def test_wide_compaction_with_many_jets():
    """1300 jets (> 1024: the single-workgroup scan walks several jets per thread), random masks incl. empty jets."""
    from particle_fm_amd import hip_ops_wide as opw
    from particle_fm_amd.layout import EpicConfig
    from particle_fm_amd.layout_wide import EpicWideLayout
    cfg = EpicConfig(num_particles=8, features=3, hidden_dim=64, latent=4, layers=1, frequencies=16, t_local_cat=True,
                     t_global_cat=True, global_cond_dim=0, local_cond_dim=0)
    wl = EpicWideLayout(cfg)
    shapes = {}
    for name, i, o in cfg.linear_shapes():
        shapes[name + ".bias"], shapes[name + ".weight_g"], shapes[name + ".weight_v"] = (o,), (o, 1), (o, i)
    st = {k: torch.from_numpy(v) for k, v in seeded_state(shapes, 9).items()}
    blob = wl.pack_blob(st).cuda()
    gen = torch.Generator().manual_seed(4)
    B, N = 1300, 8
    mask = (torch.rand(B, N, 1, generator=gen) < 0.5).float()
    x = torch.randn(B, N, 3, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    hp = dict(frequencies=16, layers=1, t_local_cat=True, t_global_cat=True, global_cond_dim=0, local_cond_dim=0, sum_scale=1e-2)
    vf = EpicVectorField(st, "", hp, freqs=wl.default_freqs())
    with torch.no_grad():
        ref = vf(t[:, None].expand(B, N), x, cond=None, mask=mask)
    v = opw.ew_forward(wl, blob, t.cuda(), x.cuda(), None, mask.cuda()).cpu()
    empty = mask.sum((1, 2)) == 0
    assert empty.any() and torch.isnan(v[empty]).all() and torch.isnan(ref[empty]).all()
    torch.testing.assert_close(v[~empty], ref[~empty], atol=2e-5, rtol=2e-4)
