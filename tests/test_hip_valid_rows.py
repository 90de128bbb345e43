"""Valid-rows-only inference of the row-matrix transformer paths (PFM_*_F_VALID_ROWS): valid particles get the numbers of the
dense evaluation (and hence the reference's), padded positions are left alone / 0."""
import pytest
import torch

from tests.conftest import load_ca_golden, load_tf_golden

pytestmark = pytest.mark.gpu


def _scattered(g, B=9, seed=4):
    gen = torch.Generator().manual_seed(seed)
    N, C = g.hp["num_particles"], g.hp["global_cond_dim"]
    mask = (torch.rand(B, N, 1, generator=gen) < 0.55).float()
    mask[:, 0] = 1.0
    mask[1] = 1.0           # a full jet
    mask[2, 1:] = 0.0       # a jet with one particle
    x = torch.randn(B, N, 3, generator=gen)
    return x, torch.randn(B, C, generator=gen), torch.rand(B, generator=gen), mask


@pytest.mark.parametrize("name", ["small", "lhco"])
def test_transformer_valid_rows_forward_and_samplers(name):
    from particle_fm_amd import hip_ops_tf as ops
    from particle_fm_amd.layout_tf import PFM_TF_F_VALID_ROWS, TfConfig, TfLayout
    g = load_tf_golden(name)
    cfg = TfConfig.from_hparams(g.hp)
    dense, comp = TfLayout(cfg), TfLayout(cfg, flags=PFM_TF_F_VALID_ROWS)
    blob = dense.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    x, cond, t, mask = _scattered(g, B=9 if name == "small" else 5)
    keep = mask.squeeze(-1) == 1
    a = ops.tf_forward(dense, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    b = ops.tf_forward(comp, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    torch.testing.assert_close(b[keep], a[keep], atol=1e-5, rtol=1e-4)
    assert torch.all(b[~keep] == 0)
    # reference vectors (prefix masks) through the compacted path
    tag = "nfe_f32/"
    xr, tr, mr, cr = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ops.tf_forward(comp, blob, tr.cuda(), xr.cuda(), cr.cuda(), mr.cuda()).cpu()
    kr = mr.squeeze(-1) == 1
    torch.testing.assert_close(v[kr], g.get(tag + "v_vec_t")[kr], atol=2e-5, rtol=2e-4)
    # samplers: valid particles follow the dense trajectory, padded ones stay at z * mask = 0
    z = x * mask
    for fn, kw in ((ops.tf_sample_midpoint, dict(ode_steps=6)), (ops.tf_sample_rk, dict(ode_steps=4, solver="rk4"))):
        da = fn(dense, blob, z.cuda(), cond.cuda(), mask.cuda(), **kw).cpu()
        db = fn(comp, blob, z.cuda(), cond.cuda(), mask.cuda(), **kw).cpu()
        torch.testing.assert_close(db[keep], da[keep], atol=1e-4, rtol=1e-3)
        assert torch.all(db[~keep] == 0)
    # no mask: nothing to compact, identical launches
    c = ops.tf_forward(comp, blob, t.cuda(), x.cuda(), cond.cuda(), None).cpu()
    d = ops.tf_forward(dense, blob, t.cuda(), x.cuda(), cond.cuda(), None).cpu()
    assert torch.equal(c, d)


@pytest.mark.parametrize("name", ["small", "lhco"])
def test_cross_attention_valid_rows_forward_and_samplers(name):
    from particle_fm_amd import hip_ops_ca as ops
    from particle_fm_amd.layout_ca import PFM_CA_F_VALID_ROWS, CaConfig, CaLayout
    g = load_ca_golden(name)
    cfg = CaConfig.from_hparams(g.hp)
    dense, comp = CaLayout(cfg), CaLayout(cfg, flags=PFM_CA_F_VALID_ROWS)
    blob = dense.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    x, cond, t, mask = _scattered(g, B=9 if name == "small" else 5)
    keep = mask.squeeze(-1) == 1
    a = ops.ca_forward(dense, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    b = ops.ca_forward(comp, blob, t.cuda(), x.cuda(), cond.cuda(), mask.cuda()).cpu()
    torch.testing.assert_close(b[keep], a[keep], atol=1e-5, rtol=1e-4)
    assert torch.all(b[~keep] == 0)
    tag = "nfe_f32/"
    xr, tr, mr, cr = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ops.ca_forward(comp, blob, tr.cuda(), xr.cuda(), cr.cuda(), mr.cuda()).cpu()
    kr = mr.squeeze(-1) == 1
    torch.testing.assert_close(v[kr], g.get(tag + "v_vec_t")[kr], atol=2e-5, rtol=2e-4)
    z = x * mask
    for fn, kw in ((ops.ca_sample_midpoint, dict(ode_steps=6)), (ops.ca_sample_rk, dict(ode_steps=4, solver="rk4"))):
        da = fn(dense, blob, z.cuda(), cond.cuda(), mask.cuda(), **kw).cpu()
        db = fn(comp, blob, z.cuda(), cond.cuda(), mask.cuda(), **kw).cpu()
        torch.testing.assert_close(db[keep], da[keep], atol=1e-4, rtol=1e-3)
        assert torch.all(db[~keep] == 0)


def test_generate_data_uses_valid_rows_and_matches_dense():
    """generate_data with variable_set_sizes multiplies by the mask at the end, so skipping padded particles cannot change its
    result: same output with the switch on (what it does by itself) and forced off."""
    import copy
    from particle_fm_amd.models import SetFlowMatchingLitModule
    from particle_fm_amd.utils.data_generation import generate_data
    g = load_tf_golden("small")
    m = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    x, cond, t, mask = _scattered(g, B=8)
    outs = []
    for allow in (True, False):
        torch.manual_seed(123)
        data, _ = generate_data(m, 8, cond=cond, batch_size=4, device="cuda", variable_set_sizes=True, mask=mask, verbose=False,
                                ode_steps=5, valid_rows_only=allow)
        outs.append(torch.from_numpy(data))
        assert m.flows[0].net.valid_rows_only is False  # restored
    torch.testing.assert_close(outs[0], outs[1], atol=1e-4, rtol=1e-3)
    assert torch.all(outs[0][mask.squeeze(-1) == 0] == 0)


@pytest.mark.parametrize("which", ["tf", "ca"])
def test_generate_data_two_stream_pipeline_changes_nothing(which):
    """generate_data's batch pipeline (weights packed once, batches alternating between two streams, graph replay of the
    cross-attention step) for the row-matrix models: the same array as one batch after the other."""
    import copy
    from particle_fm_amd.models import SetFlowMatchingLitModule
    from particle_fm_amd.utils.data_generation import generate_data
    from tests.conftest import load_ca_golden
    g = load_tf_golden("small") if which == "tf" else load_ca_golden("small")
    m = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    N, C = g.hp["num_particles"], g.hp["global_cond_dim"]
    gen = torch.Generator().manual_seed(5)
    n = torch.randint(1, N + 1, (20,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    cond = torch.randn(20, C, generator=gen)
    outs = []
    for pipe in (True, False):
        torch.manual_seed(321)
        data, _ = generate_data(m, 20, cond=cond, batch_size=6, device="cuda", variable_set_sizes=True, mask=mask, verbose=False,
                                ode_steps=6, pipeline=pipe)
        outs.append(torch.from_numpy(data))
        assert getattr(m.flows[0].net, "graph_replay", False) is False  # restored
    assert torch.equal(outs[0], outs[1])
    assert torch.all(outs[0][mask.squeeze(-1) == 0] == 0)
