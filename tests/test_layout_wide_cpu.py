"""layout_wide.py: the padded MFMA_AK blob of the wide EPiC path, decoded and evaluated on the CPU along the kernels'
own dataflow (P / Q rows, jet-bias GEMMs), against the reference vectors."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_layout_cpu import cfg_of
from tests.tf_blob_interp import mfma_ak


def _layout(g):
    from particle_fm_amd.layout_wide import EpicWideLayout
    return EpicWideLayout(cfg_of(g.hp))


def _lin(blob, lin, NO, K, x, check_T=True):
    W = mfma_ak(blob, lin.W, NO, K)
    if check_T and lin.WT >= 0:
        assert torch.equal(mfma_ak(blob, lin.WT, K, NO), W.t())
    y = x @ W.t()
    return y + blob[lin.b:lin.b + NO] if lin.b >= 0 else y


def interp_forward(d, blob, t, x, cond, mask):
    B, N, Fe = x.shape
    Hp, T, C, s = d.hidden_pad, d.t_dim, d.cond_global, d.neg_slope
    act = lambda a: F.leaky_relu(a, s)
    P = torch.zeros(B, 256 + Hp)
    fr = blob[d.freqs:d.freqs + T]
    if d.flags & 2:  # PFM_EW_F_TEMB_SINCOS: table = [f ; f]
        a = fr * t[:, None]
        P[:, :T] = torch.cat([a[:, :T // 2].cos(), a[:, T // 2:].sin()], -1)
    else:
        P[:, :T] = torch.cos((t[:, None] + 0.0) * fr * math.pi / 1.0)
    if C:
        P[:, T:T + C] = cond
    m = torch.ones(B, N) if mask is None else mask.reshape(B, N).float()
    sjb = _lin(blob, d.sjb, 2 * Hp + 128, 256, P[:, :256])
    Wx = blob[d.l1x:d.l1x + Fe * Hp].reshape(Fe, Hp)
    x1 = act(x @ Wx + sjb[:, None, :Hp])
    X = act(_lin(blob, d.l2, Hp, Hp, x1) + sjb[:, None, Hp:2 * Hp] + x1)

    def pool():
        sm = (X * m[..., None]).sum(1)
        return torch.cat([sm / m.sum(1, keepdim=True), sm * d.sum_scale], -1)

    P[:, 256:] = act(_lin(blob, d.sg1, Hp, 256 + 2 * Hp, torch.cat([P[:, :256], pool()], -1)))
    P[:, 128:256] = act(_lin(blob, d.sg2, 128, 256 + Hp, P))
    for l in range(d.layers):
        L = d.layer[l]
        P[:, 256:] = act(_lin(blob, L.g1, Hp, 256 + 2 * Hp, torch.cat([P[:, :256], pool()], -1)))
        P[:, 128:256] = act(_lin(blob, L.g2, 128, 256 + Hp, P) + P[:, 128:256])
        jb = _lin(blob, L.jb, 2 * Hp, 256, P[:, :256])
        l1 = act(_lin(blob, L.l1, Hp, Hp, X) + jb[:, None, :Hp])
        X = act(_lin(blob, L.l2, Hp, Hp, l1) + jb[:, None, Hp:] + X)
    assert torch.all(X[..., d.hidden:] == 0), "padding columns must stay exactly zero"
    W3 = blob[d.l3:d.l3 + Fe * Hp].reshape(Fe, Hp)
    return act(X @ W3.t() + sjb[:, None, 2 * Hp:2 * Hp + Fe]) * m[..., None]


@pytest.mark.parametrize("name", ["small", "sincos"])
def test_blob_evaluates_to_reference(name):
    from tests.conftest import load_wide_golden
    g = load_wide_golden(name)
    lay = _layout(g)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs)
    assert blob.numel() == lay.blob_total
    for mk in ("f32", "none"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        v = interp_forward(lay.desc, blob, t, x, cond, mask)
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), rtol=1e-4, atol=2e-5)


def test_grad_pos_covers_every_weight_once():
    from tests.conftest import load_wide_golden
    lay = _layout(load_wide_golden("small"))
    n = lay.freq_off
    assert lay.grad_pos.shape == (n,) and len(np.unique(lay.grad_pos)) == n
    assert np.array_equal(lay.index_map[lay.grad_pos], np.arange(n))


def test_also_packs_the_narrow_configs(golden):
    """hidden 128 goes through the same layout (Hp = 128): the wide path is a superset of the jet-resident one."""
    lay = _layout(golden)
    blob = lay.pack_blob(golden.state, "flows.0.net.", freqs=golden.freqs)
    tag = "nfe_f32/"
    x, t, mask, cond = (golden.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = interp_forward(lay.desc, blob, t, x, cond, mask)
    torch.testing.assert_close(v, golden.get(tag + "v_vec_t"), rtol=1e-4, atol=1e-5)
