"""The cross-attention drop-in modules on the GPU: reference call signatures in, reference numbers out."""
import copy

import pytest
import torch

from oracle.ca_ref import CrossAttentionVectorField
from oracle.fm_ref import fm_ot_loss, sample_midpoint

pytestmark = pytest.mark.gpu


def _module(g, **over):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    kw = copy.deepcopy(g.hp)
    kw.update(over)
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **kw)
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    return m.cuda()


def _oracle(g, state=None):
    from particle_fm_amd.layout_ca import default_freqs
    # the product fixes the frequency table; give the oracle the same one
    return CrossAttentionVectorField(state or g.state, "flows.0.", g.hp, freqs=default_freqs(2 * g.hp["frequencies"], g.hp.get("t_emb", "cosine")))


def test_cnf_forward_reference_signature(ca_golden):
    g = ca_golden
    m = _module(g)
    tag = "nfe_int64/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    N = x.shape[1]
    vf = _oracle(g)
    with torch.no_grad():
        ref = vf(t[:, None].expand(-1, N), x, cond=cond, mask=mask)
        ref_s = vf(t[0], x, cond=cond, mask=mask)
    tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)  # losses.py:47 shape (B,N)
    v = m.flows[0](tt.cuda(), x.cuda(), cond=cond.cuda(), mask=mask.cuda()).cpu()
    torch.testing.assert_close(v, ref, atol=2e-5, rtol=2e-4)
    vs = m.flows[0](t[0].cuda(), x.cuda(), cond=cond.cuda(), mask=mask.cuda()).cpu()  # 0-dim t (sampling)
    torch.testing.assert_close(vs, ref_s, atol=2e-5, rtol=2e-4)
    # split-fp16 Linears: fp32-grade agreement with the fp32 path
    m.flows[0].net.set_precision("f16x3")
    v3 = m.flows[0](tt.cuda(), x.cuda(), cond=cond.cuda(), mask=mask.cuda()).cpu()
    torch.testing.assert_close(v3, ref, atol=2e-5, rtol=2e-4)


def test_sample_matches_oracle():
    from tests.conftest import load_ca_golden
    g = load_ca_golden("small")
    m = _module(g)
    tag = "midpoint_10/"
    mask, cond = g.get(tag + "mask"), g.get(tag + "cond")
    B, N, F = mask.shape[0], g.hp["num_particles"], g.hp["features"]
    torch.manual_seed(9999)
    out = m.sample(B, cond=cond, mask=mask, ode_solver="midpoint", ode_steps=20).cpu()
    torch.manual_seed(9999)
    z = torch.randn(B, N, F)  # flow_matching_module.py:659-663 draws on the CPU generator
    ref = sample_midpoint(_oracle(g), z, cond, mask, ode_steps=20)
    torch.testing.assert_close(out, ref, atol=2e-4, rtol=1e-3)


def test_training_step_and_optimizer_step(ca_golden):
    """training_step with the reference's draws replayed, gradients on the real nn.Parameters, then one fused
    clip + AdamW + EMA step against torch's own on the oracle."""
    from particle_fm_amd.engine import FusedFMTrainer
    g = ca_golden
    m = _module(g)
    tag = "loss_f32/"
    x, mask, cond = (g.get(tag + k) for k in ("x", "mask", "cond"))
    tr = FusedFMTrainer(m, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999)
    torch.manual_seed(4321)
    loss = tr.step((x.cuda(), mask.cuda(), cond.cuda()))
    torch.manual_seed(4321)
    t = torch.rand_like(torch.ones(x.shape[0]))  # losses.py:46 (CPU generator)
    z = torch.randn_like(x.cuda()).cpu()         # losses.py:53 (device generator)
    ref = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if "frequencies" not in k}
    params = list(ref.values())
    l_ref, *_ = fm_ot_loss(_oracle(g, ref), x, mask, cond, t, z, sigma=1e-4)
    l_ref.backward()
    torch.testing.assert_close(loss.cpu(), l_ref.detach(), atol=2e-6, rtol=2e-5)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=5e-5)
    gn = torch.nn.utils.clip_grad_norm_(params, 0.5)
    torch.testing.assert_close(tr.grad_norm().cpu(), gn, atol=1e-5, rtol=1e-3)
    opt.step()
    got = dict(m.named_parameters())
    for k, p in ref.items():
        if k.endswith("k_linear.bias"):
            continue  # gradient 0 in exact arithmetic: Adam's first step amplifies pure rounding noise to +-lr
        upd_ref = p.detach() - g.state[k]
        upd = got[k].detach().cpu() - g.state[k]
        bad = int(((upd - upd_ref).abs() > 1e-5).sum())
        assert bad <= max(2, 2e-3 * upd.numel()), (k, bad, upd.numel())
