"""oracle/diffusion_ref.py against vectors recorded from the reference's DiffusionLoss, ode_wrapper (diffusion branch),
ddim_sampler and euler_maruyama_sampler (tests/golden/epic_diffusion.npz)."""
import pytest
import torch

from oracle import diffusion_ref as dr
from oracle.fm_ref import EpicVectorField, midpoint_trajectory_end
from tests.conftest import load_golden


def _vf(g, state=None):
    return EpicVectorField(state or g.state, "flows.0.net", g.hp, freqs=g.freqs)


@pytest.mark.parametrize("crit", ["huber", "mse"])
def test_loss_and_gradients(crit):
    g = load_golden("diffusion")
    state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
    tag = f"loss_{crit}/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    loss, noisy, pred = dr.diffusion_loss(_vf(g, state), x, mask, cond, t, z, crit, g.hp["diff_config"])
    if crit == "huber":
        assert float(((z - pred).abs() * mask > 1).float().mean()) > 0.01  # both branches of the huber criterion are exercised
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
    loss.backward()
    ref = g.grads(tag)
    assert len(ref) == len(state)
    for k, want in ref.items():
        assert float((state[k].grad - want).norm()) <= 2e-4 * float(want.norm()) + 1e-7, k


def test_ode_rhs_and_midpoint():
    g = load_golden("diffusion")
    vf, dc = _vf(g), g.hp["diff_config"]
    t, x, mask, cond = (g.get("rhs/" + k) for k in ("t", "x", "mask", "cond"))
    with torch.no_grad():
        f = dr.diffusion_rhs(vf, t, x, cond, mask, dc)
    torch.testing.assert_close(f, g.get("rhs/f"), rtol=1e-5, atol=1e-5)
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        with torch.no_grad():
            xe = midpoint_trajectory_end(lambda tt, xx: dr.diffusion_rhs(vf, tt, xx, cond, mask, dc), z * mask, torch.linspace(1.0, 0.0, steps))
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-4, atol=5e-5)


def test_ddim_and_euler_maruyama():
    g = load_golden("diffusion")
    vf, dc, n = _vf(g), g.hp["diff_config"], int(g.z["n_steps"])
    z, mask, cond = (g.get("ddim/" + k) for k in ("z", "mask", "cond"))
    torch.testing.assert_close(dr.ddim_sample(vf, z * mask, cond, mask, n, dc), g.get("ddim/x_end"), rtol=1e-4, atol=5e-5)
    z, mask, cond, noise = (g.get("em/" + k) for k in ("z", "mask", "cond", "noise"))
    torch.testing.assert_close(dr.em_sample(vf, z * mask, cond, mask, n, dc, noise), g.get("em/x_end"), rtol=1e-4, atol=5e-5)


# ---- the same recordings for the Full-Transformer / cross-attention / MDMA models (tests/golden/{tf,ca,mdma}_diffusion.npz) ----
def _rows(path):
    """path: "tf" / "ca" / "mdma", or with "_gauss" the t_emb="gaussian" recordings (<model>_diffusion_gauss.npz; there also "epic" -- the
    jet-resident configuration -- and "epicw")."""
    from tests.conftest import load_ca_golden, load_epic_seeded_golden, load_mdma_golden, load_tf_golden, load_wide_golden
    base, _, gauss = path.partition("_")
    loader = {"tf": load_tf_golden, "ca": load_ca_golden, "mdma": load_mdma_golden, "epic": load_epic_seeded_golden, "epicw": load_wide_golden}[base]
    g = loader("diffusion_gauss" if gauss else "diffusion")
    path = base
    if path in ("epic", "epicw"):
        from oracle.fm_ref import EpicVectorField
        return g, EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    if path == "tf":
        from oracle.tf_ref import TransformerVectorField as VF
    elif path == "ca":
        from oracle.ca_ref import CrossAttentionVectorField as VF
    else:
        from oracle.mdma_ref import MdmaVectorField
        base = MdmaVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)
        return g, (lambda t, x, mask=None, cond=None: base(t, x, cond, mask))
    return g, VF(g.state, "flows.0.", g.hp, freqs=g.freqs)


def _c(g, tag):
    c = g.get(tag + "cond")
    return None if c is None or c.numel() == 0 else c


@pytest.mark.parametrize("path", ["tf", "ca", "mdma", "tf_gauss", "ca_gauss", "mdma_gauss", "epic_gauss", "epicw_gauss"])
def test_row_models_loss_rhs_and_samplers(path):
    g, vf = _rows(path)
    dc, n = g.hp["diff_config"], int(g.z["n_steps"])
    for crit in ("huber", "mse"):
        tag = f"loss_{crit}/"
        x, t, z, mask = (g.get(tag + k) for k in ("x", "t", "z", "mask"))
        with torch.no_grad():
            loss, *_ = dr.diffusion_loss(vf, x, mask, _c(g, tag), t, z, crit, dc)
        torch.testing.assert_close(loss, g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
    t, x, mask = (g.get("rhs/" + k) for k in ("t", "x", "mask"))
    with torch.no_grad():
        f = dr.diffusion_rhs(vf, t, x, _c(g, "rhs/"), mask, dc)
        torch.testing.assert_close(f.expand_as(g.get("rhs/f")), g.get("rhs/f"), rtol=1e-4, atol=1e-4)
        for steps in (3, 10):
            tag = f"midpoint_{steps}/"
            z, mask = g.get(tag + "z"), g.get(tag + "mask")
            cond = _c(g, tag)
            xe = midpoint_trajectory_end(lambda tt, xx: dr.diffusion_rhs(vf, tt, xx, cond, mask, dc), z * mask, torch.linspace(1.0, 0.0, steps))
            torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=2e-4)
        z, mask = g.get("ddim/z"), g.get("ddim/mask")
        torch.testing.assert_close(dr.ddim_sample(vf, z * mask, _c(g, "ddim/"), mask, n, dc), g.get("ddim/x_end"), rtol=1e-3, atol=2e-4)


def test_chained_diffusion_fixture_is_reproduced_by_the_oracle():
    """tests/golden/epic_chain2_diffusion.npz (n_transforms = 2 under DiffusionLoss, losses.py:264-267): the oracle's fields composed the
    same way reproduce the recorded losses and the reverse-order midpoint samples."""
    from oracle.fm_ref import EpicVectorField
    from tests.conftest import load_epic_seeded_golden
    g = load_epic_seeded_golden("chain2_diffusion")
    dc = g.hp["diff_config"]
    vfs = [EpicVectorField(g.state, f"flows.{i}.net", g.hp, freqs=g.freqs) for i in range(2)]

    def chain(t, x, mask=None, cond=None):
        for vf in vfs:
            x = vf(t, x, mask=mask, cond=cond)
        return x

    with torch.no_grad():
        for crit in ("huber", "mse"):
            tag = f"loss_{crit}/"
            x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "a", "mask", "cond"))
            loss, *_ = dr.diffusion_loss(chain, x, mask, cond, t, z, crit, dc)
            torch.testing.assert_close(loss, g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
        tag = "midpoint_10/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        xe = z * mask
        for vf in reversed(vfs):
            xe = midpoint_trajectory_end(lambda tt, xx: dr.diffusion_rhs(vf, tt, xx, cond, mask, dc), xe, torch.linspace(1.0, 0.0, 10))
        torch.testing.assert_close(xe, g.get(tag + "x_end"), rtol=1e-3, atol=2e-4)
