"""t_emb="gaussian" (SURVEY 8f-3): the oracle's restatement of the learned time embedding (flow_matching_module.py:178-181,
213-221; time_emb.py:9-22) against vectors recorded from the reference's own CNF (tests/golden/epic_gauss.npz)."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, fm_ot_loss, gaussian_time_embedding, sample_midpoint
from tests.conftest import load_golden


@pytest.fixture(scope="module")
def g():
    return load_golden("gauss")


def test_fixture_is_the_gaussian_configuration(g):
    assert g.hp["t_emb"] == "gaussian"
    for k, shp in (("embed.0.W", (64,)), ("embed.1.weight", (128, 128)), ("embed.1.bias", (128,)), ("linear.weight", (32, 128)),
                   ("linear.bias", (32,))):
        assert tuple(g.state["flows.0." + k].shape) == shp


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_embedding_and_nfe(g, mk):
    tag = f"nfe_{mk}/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    N = x.shape[1]
    tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
    with torch.no_grad():
        temb = gaussian_time_embedding(tt, x, g.state, "flows.0.")
        torch.testing.assert_close(temb[:, 0, :], g.get(tag + "temb"), atol=1e-6, rtol=1e-6)
        vf = EpicVectorField(g.state, "flows.0.net", g.hp)
        torch.testing.assert_close(vf(tt, x, cond=cond, mask=mask), g.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(vf(t[0], x, cond=cond, mask=mask), g.get(tag + "v_scalar_t"), atol=1e-5, rtol=1e-4)


def test_loss_and_gradients_including_the_embedding_parameters(g):
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k and not k.endswith("embed.0.W"))
          for k, v in g.state.items()}
    vf = EpicVectorField(st, "flows.0.net", g.hp)
    loss, *_ = fm_ot_loss(vf, x, mask, cond, t, z, sigma=1e-4)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), atol=1e-6, rtol=1e-5)
    loss.backward()
    ref = g.grads(tag)
    assert {"flows.0.embed.1.weight", "flows.0.embed.1.bias", "flows.0.linear.weight", "flows.0.linear.bias"} <= set(ref)
    for k, gref in ref.items():
        scale = max(gref.abs().max().item(), 1e-8)
        assert (st[k].grad - gref).abs().max().item() / scale <= 2e-5, k


@pytest.mark.parametrize("steps", [3, 10])
def test_midpoint(g, steps):
    tag = f"midpoint_{steps}/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    vf = EpicVectorField(g.state, "flows.0.net", g.hp)
    torch.testing.assert_close(sample_midpoint(vf, z, cond, mask, ode_steps=steps), g.get(tag + "x_end"), atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("path", ["tf", "ca", "mdma"])
def test_gaussian_embedding_on_the_transformer_fields(path):
    """tests/golden/{tf,ca,mdma}_gauss.npz: the oracle's transformer / cross-attention / MDMA fields (the latter with every time
    concatenation on) with the CNF's gaussian time-embedding network reproduce the reference's recorded forward vectors, FM-OT loss and
    midpoint samples."""
    from oracle.fm_ref import fm_ot_loss, midpoint_trajectory_end
    from tests.conftest import load_ca_golden, load_mdma_golden, load_tf_golden
    if path == "tf":
        from oracle.tf_ref import TransformerVectorField as VF
        g = load_tf_golden("gauss")
    elif path == "ca":
        from oracle.ca_ref import CrossAttentionVectorField as VF
        g = load_ca_golden("gauss")
    else:
        from oracle.mdma_ref import MdmaVectorField as VF
        g = load_mdma_golden("gauss")
    vf = VF(g.state, "flows.0.", g.hp, freqs=g.freqs)
    with torch.no_grad():
        tag = "nfe_f32/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        v = vf(t[:, None].expand(-1, x.shape[1]), x, cond=cond, mask=mask)
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)
        tag = "loss_f32/"
        x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
        loss, *_ = fm_ot_loss(vf, x, mask, cond, t, z, sigma=1e-4)
        torch.testing.assert_close(loss, g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
        tag = "midpoint_10/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        xe = midpoint_trajectory_end(lambda tt, xx: vf(tt, xx, mask=mask, cond=cond), z * mask, torch.linspace(1.0, 0.0, 10))
        torch.testing.assert_close(xe, g.get(tag + "x_end"), atol=5e-5, rtol=1e-3)


def test_gaussian_embedding_on_the_wide_epic_field():
    """tests/golden/epicw_gauss.npz (hidden 300: the row-matrix path's width): the oracle's EPiC field behind the CNF's gaussian
    embedding network reproduces the reference's recorded forward vectors, FM-OT loss + gradients and midpoint samples."""
    from tests.conftest import load_wide_golden
    g = load_wide_golden("gauss")
    assert g.hp["t_emb"] == "gaussian" and g.hp["hidden_dim"] == 300
    vf = EpicVectorField(g.state, "flows.0.net", g.hp)
    with torch.no_grad():
        for mk in ("f32", "none"):
            tag = f"nfe_{mk}/"
            x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
            v = vf(t[:, None].expand(-1, x.shape[1]), x, cond=cond, mask=mask)
            torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)
            torch.testing.assert_close(vf(t[0], x, cond=cond, mask=mask), g.get(tag + "v_scalar_t"), atol=1e-5, rtol=1e-4)
        tag = "midpoint_10/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        torch.testing.assert_close(sample_midpoint(vf, z, cond, mask, ode_steps=10), g.get(tag + "x_end"), atol=5e-5, rtol=1e-3)
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k and not k.endswith("embed.0.W"))
          for k, v in g.state.items()}
    loss, *_ = fm_ot_loss(EpicVectorField(st, "flows.0.net", g.hp), x, mask, cond, t, z, sigma=1e-4)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), atol=1e-6, rtol=1e-5)
    loss.backward()
    ref = g.grads(tag)
    assert {"flows.0.embed.1.weight", "flows.0.embed.1.bias", "flows.0.linear.weight", "flows.0.linear.bias"} <= set(ref)
    for k, gref in ref.items():
        scale = max(gref.abs().max().item(), 1e-8)
        assert (g.pick(st[k].grad) - gref).abs().max().item() / scale <= 5e-5, k
