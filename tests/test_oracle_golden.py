"""The CPU oracle (oracle/*.py) against vectors recorded from the REFERENCE's own modules
(tests/golden/*.npz, written by oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle.epic_ref import epic_param_shapes, wn_linear
from oracle.fm_ref import (
    EpicVectorField,
    cfm_loss,
    cosine_encoding,
    fm_ot_loss,
    midpoint_time_grid,
    sample_midpoint,
)

# fp32 NFE tolerance (fp32-vs-fp64 floor 2.3e-7 on |v|~0.12)
ATOL, RTOL = 1e-5, 1e-4


def vf_of(g):
    return EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)


def test_state_dict_keys_and_shapes(golden):
    hp = golden.hp
    T = 2 * hp["frequencies"]
    expect = epic_param_shapes(
        features=hp["features"], input_dim=hp["features"] + (T if hp.get("add_time_to_input") else 0), hidden=hp["hidden_dim"],
        latent=hp["latent"], layers=hp["layers"], t_dim_local=T if hp["t_local_cat"] else 0, t_dim_global=T if hp["t_global_cat"] else 0,
        global_cond_dim=hp["global_cond_dim"], local_cond_dim=hp["local_cond_dim"],
    )
    keys = ["flows.0.frequencies"] + ["flows.0.net." + k for k, _ in expect]
    assert golden.keys == keys
    for k, shp in expect:
        assert tuple(golden.state["flows.0.net." + k].shape) == shp


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_time_embedding_bitwise(golden, mk):
    t = golden.get(f"nfe_{mk}/t")
    temb = cosine_encoding(t, 2 * golden.hp["frequencies"], freqs=golden.freqs)
    # same torch CPU kernels, same op order -> bit-identical
    assert torch.equal(temb, golden.get(f"nfe_{mk}/temb"))


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_nfe_vector_and_scalar_t(golden, mk):
    tag = f"nfe_{mk}/"
    x, t = golden.get(tag + "x"), golden.get(tag + "t")
    mask, cond = golden.get(tag + "mask"), golden.get(tag + "cond")
    vf = vf_of(golden)
    N = x.shape[1]
    with torch.no_grad():
        v = vf(t.unsqueeze(-1).repeat_interleave(N, dim=1), x, cond=cond, mask=mask)
        vs = vf(t[0], x, cond=cond, mask=mask)
    torch.testing.assert_close(v, golden.get(tag + "v_vec_t"), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(vs, golden.get(tag + "v_scalar_t"), atol=ATOL, rtol=RTOL)
    # scalar-t call == vector-t call on the jet whose t it took
    torch.testing.assert_close(vs[0], v[0], atol=ATOL, rtol=RTOL)
    if mask is not None:
        assert torch.all(v[mask.squeeze(-1) == 0] == 0)


@pytest.mark.parametrize("mk", ["f32", "none"])
def test_fm_loss_and_grads(golden, mk):
    tag = f"loss_{mk}/"
    x, t, z = golden.get(tag + "x"), golden.get(tag + "t"), golden.get(tag + "z")
    mask, cond = golden.get(tag + "mask"), golden.get(tag + "cond")
    state = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in golden.state.items()}
    vf = EpicVectorField(state, "flows.0.net", golden.hp, freqs=golden.freqs)
    loss, *_ = fm_ot_loss(vf, x, mask, cond, t, z, sigma=1e-4)
    torch.testing.assert_close(loss.detach(), golden.get(tag + "loss"), atol=1e-6, rtol=1e-5)
    loss.backward()
    ref = golden.grads(tag)
    assert len(ref) == len(golden.keys) - 1
    for k, gref in ref.items():
        got = state[k].grad
        scale = max(gref.abs().max().item(), 1e-8)
        assert (got - gref).abs().max().item() <= 2e-5 * scale + 1e-7, k


def test_cfm_loss(golden):
    tag = "cfm/"
    vf = vf_of(golden)
    with torch.no_grad():
        loss, *_ = cfm_loss(vf, golden.get(tag + "x"), golden.get(tag + "mask"), golden.get(tag + "cond"),
                            golden.get(tag + "t"), golden.get(tag + "x0"), golden.get(tag + "eps"), sigma=1e-4)
    torch.testing.assert_close(loss, golden.get(tag + "loss"), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("steps", [3, 10, 100])
def test_midpoint_reference_vf_restated_integrator(golden, steps):
    tag = f"midpoint_{steps}/"
    vf = vf_of(golden)
    xe = sample_midpoint(vf, golden.get(tag + "z"), golden.get(tag + "cond"), golden.get(tag + "mask"), ode_steps=steps)
    # fp32 noise floor after 99 steps is 2.7e-6 abs on |x|~3.9
    torch.testing.assert_close(xe, golden.get(tag + "x_end"), atol=5e-5, rtol=1e-4)


def test_midpoint_grid_matches_driver():
    ts, dts = midpoint_time_grid(100)
    assert ts.shape == (198,) and dts.shape == (99,)
    assert ts[0].item() == 1.0
    assert abs(ts[-1].item() - (1.0 / 99) * 0.5) < 1e-6


def test_weight_norm_rowwise():
    torch.manual_seed(0)
    lin = torch.nn.utils.weight_norm(torch.nn.Linear(7, 5))
    with torch.no_grad():
        lin.weight_g.mul_(1.3)
    st = {"l." + k: v.detach() for k, v in lin.state_dict().items()}
    x = torch.randn(3, 7)
    torch.testing.assert_close(wn_linear(st, "l", x), lin(x).detach())


@pytest.mark.parametrize("fixture", ["chain2", "chain2w"])
def test_chained_flows_fixture_is_reproduced_by_the_oracle(fixture):
    """tests/golden/epic_chain2.npz, epic_chain2w.npz (n_transforms = 2, flow_matching_module.py:421-443; hidden 128 / 136): the oracle's
    field composed as the reference's losses compose it (losses.py:66-69) reproduces the recorded FM-OT loss and the reverse-order
    midpoint samples."""
    from oracle.fm_ref import EpicVectorField, fm_ot_targets, midpoint_trajectory_end
    from tests.conftest import load_golden
    g = load_golden(fixture)
    vfs = [EpicVectorField(g.state, f"flows.{i}.net", g.hp, freqs=g.freqs) for i in range(2)]
    tag = "loss_fm/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "a", "mask", "cond"))
    with torch.no_grad():
        tt, y, u, m = fm_ot_targets(x, mask, t, z, 1e-4)
        temp = y
        for vf in vfs:
            temp = vf(tt.squeeze(-1), temp, mask=m, cond=cond)
        loss = (temp - u).square().sum() / m.sum()
        torch.testing.assert_close(loss, g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
        for steps in (3, 10):
            tag = f"midpoint_{steps}/"
            z, mk, c = (g.get(tag + k) for k in ("z", "mask", "cond"))
            xe = z * mk
            for vf in reversed(vfs):
                xe = midpoint_trajectory_end(lambda tq, xx: vf(tq, xx, mask=mk, cond=c), xe, torch.linspace(1.0, 0.0, steps))
            torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("path", ["tf", "ca", "epic_gauss", "epicw_gauss"])
def test_chained_transformer_flows_fixture_is_reproduced_by_the_oracle(path):
    """tests/golden/{tf,ca}_chain2.npz (n_transforms = 2 on the Full-Transformer / cross-attention models): the oracle's fields composed as
    losses.py:66-69 composes them reproduce the recorded FM-OT loss and the reverse-order midpoint samples."""
    from oracle.fm_ref import fm_ot_targets, midpoint_trajectory_end
    from tests.conftest import load_ca_golden, load_epic_seeded_golden, load_tf_golden, load_wide_golden
    if path == "tf":
        from oracle.tf_ref import TransformerVectorField as VF
        g = load_tf_golden("chain2")
    elif path == "ca":
        from oracle.ca_ref import CrossAttentionVectorField as VF
        g = load_ca_golden("chain2")
    else:  # both EPiC paths with t_emb="gaussian"
        from oracle.fm_ref import EpicVectorField
        g = (load_epic_seeded_golden if path == "epic_gauss" else load_wide_golden)("chain2_gauss")
        VF = lambda st, pre, hp, freqs=None: EpicVectorField(st, pre + "net", hp, freqs=freqs)
    vfs = [VF(g.state, f"flows.{i}.", g.hp, freqs=g.freqs) for i in range(2)]
    tag = "loss_fm/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "a", "mask", "cond"))
    with torch.no_grad():
        tt, y, u, m = fm_ot_targets(x, mask, t, z, 1e-4)
        temp = y
        for vf in vfs:
            temp = vf(tt.squeeze(-1), temp, mask=m, cond=cond)
        loss = (temp - u).square().sum() / m.sum()
        torch.testing.assert_close(loss, g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
        tag = "midpoint_10/"
        z, mk, c = (g.get(tag + k) for k in ("z", "mask", "cond"))
        xe = z * mk
        for vf in reversed(vfs):
            xe = midpoint_trajectory_end(lambda tq, xx: vf(tq, xx, mask=mk, cond=c), xe, torch.linspace(1.0, 0.0, 10))
        keep = mk.squeeze(-1) != 0
        torch.testing.assert_close(xe[keep], g.get(tag + "x_end")[keep], rtol=1e-3, atol=5e-5)
