"""IterativeNormLayer mirror (use_normaliser=True) on the GPU against the reference's recorded statistics and outputs."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "norm_layer.npz"))


def t(k):
    return torch.from_numpy(G[k])


def test_layer_matches_reference_vectors():
    from particle_fm_amd.models.components.norm_layer import IterativeNormLayer
    layer = IterativeNormLayer((3,), max_n=int(G["max_n"])).cuda()
    layer.train()
    assert list(layer.state_dict().keys()) == ["means", "vars", "n", "m2"]
    for k in range(4):
        tag = f"step{k}/"
        y = layer(t(tag + "x").cuda(), t(tag + "mask").cuda())
        torch.testing.assert_close(y.cpu(), t(tag + "y"), rtol=2e-5, atol=2e-6)
        for b in ("means", "vars", "m2"):
            torch.testing.assert_close(getattr(layer, b).cpu(), t(tag + b), rtol=2e-5, atol=2e-6)
        assert int(layer.n) == int(G[tag + "n"])  # incl. no change once n >= max_n (step 3)
    layer.eval()
    rev = layer.reverse(t("step3/y").cuda(), t("step3/mask").cuda()).cpu()
    torch.testing.assert_close(rev, t("rev/y"), rtol=2e-5, atol=2e-6)
    cl = IterativeNormLayer((2,), max_n=250).cuda().train()
    torch.testing.assert_close(cl(t("cond/x").cuda()).cpu(), t("cond/y"), rtol=2e-5, atol=2e-6)
    # fit(): population statistics, then frozen
    fl = IterativeNormLayer((3,)).cuda()
    fl.fit(t("step0/x").cuda(), t("step0/mask").cuda())
    torch.testing.assert_close(fl.means.cpu(), t("step0/means"), rtol=2e-5, atol=2e-6)
    assert fl.frozen and int(fl.n) == int(G["step0/n"])


def test_lit_module_with_normaliser():
    """training_step normalises x (and cond) before the loss; sample() un-normalises the generated particles."""
    from oracle import norm_ref
    from oracle.fm_ref import EpicVectorField, fm_ot_loss, sample_midpoint
    from particle_fm_amd.layout import EpicLayout
    from particle_fm_amd.models import SetFlowMatchingLitModule
    from tests.conftest import load_golden
    from tests.test_layout_cpu import cfg_of
    g = load_golden("cond_gl")
    hp = copy.deepcopy(g.hp)
    m = SetFlowMatchingLitModule(optimizer=None, use_normaliser=True, normaliser_config={"max_n": 2000}, **hp)
    sd = m.state_dict()
    assert {"normaliser.means", "normaliser.vars", "normaliser.n", "normaliser.m2", "ctxt_normaliser.means"} <= set(sd)
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full, strict=False)
    m = m.cuda().train()
    tag = "loss_f32/"
    x, mask, cond = (g.get(tag + k) for k in ("x", "mask", "cond"))
    torch.manual_seed(11)
    loss = m.training_step((x.cuda(), mask.cuda(), cond.cuda()), 0)["loss"]
    # oracle: same pre-processing, same draws
    bm = mask.squeeze(-1) == 1
    st, sc = norm_ref.new_state(x.shape[-1]), norm_ref.new_state(cond.shape[-1])
    norm_ref.update(st, x, bm)
    norm_ref.update(sc, cond)
    xn, cn = norm_ref.forward(st, x, bm), norm_ref.forward(sc, cond)
    torch.manual_seed(11)
    tt = torch.rand_like(torch.ones(x.shape[0]))
    z = torch.randn_like(x.cuda()).cpu()
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=EpicLayout(cfg_of(g.hp)).default_freqs())
    ref, *_ = fm_ot_loss(vf, xn, mask, cn, tt, z, sigma=m.hparams.sigma)
    torch.testing.assert_close(loss.detach().cpu(), ref, rtol=5e-5, atol=1e-6)
    m.eval()
    torch.manual_seed(9999)
    out = m.sample(x.shape[0], cond=cond, mask=mask, ode_solver="midpoint", ode_steps=10).cpu()
    torch.manual_seed(9999)
    zz = torch.randn(x.shape[0], x.shape[1], x.shape[2])
    gen = sample_midpoint(vf, zz, norm_ref.forward(sc, cond), mask, ode_steps=10)
    torch.testing.assert_close(out, norm_ref.reverse(st, gen, bm), rtol=1e-3, atol=2e-4)
