"""Bit-reproducible gradients on the row-matrix paths: the Full-Transformer (cfg 4), the wide EPiC path
(cfg 5), the cross-attention encoder and MDMA used fp32 atomics for LayerNorm gamma / beta, bias and column sums and for the two
loss sums -- the order of the additions, hence the last bits, changed from run to run.  They are per-workgroup partial sums added
in block order now (csrc/tf_bwd.h: tf_ordered_sum_kernel; the loss in one workgroup), like the jet-resident path's dW: the same
inputs give the same bits.  Batches are tiled from the fixtures to a few thousand rows so that hundreds of workgroups take part."""
import pytest
import torch

pytestmark = pytest.mark.gpu

REPS = 3


def _tile(tensors, times):
    return [None if a is None else a.repeat((times,) + (1,) * (a.dim() - 1)).contiguous() for a in tensors]


def _same(runs):
    (l0, g0), rest = runs[0], runs[1:]
    assert torch.isfinite(l0).all() and torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    for l, g in rest:
        assert torch.equal(l, l0), f"loss differs between two runs: {float(l)} vs {float(l0)}"
        assert torch.equal(g, g0), f"gradient differs between two runs (max {float((g - g0).abs().max()):.3e})"


def test_full_transformer_gradients_are_bitwise_repeatable():
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    from tests.conftest import load_tf_golden
    g = load_tf_golden("lhco")  # BASELINE cfg 4 shapes: N = 279, D = 256, 3 layers
    lay = TfLayout(TfConfig.from_hparams(g.hp))
    tag = "loss_f32/"
    x, t, mask, cond, z = _tile([g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond", "z")], 12)  # 24 jets = 6696 rows
    runs = []
    for _ in range(REPS):
        flat = torch.cat([g.state[k].reshape(-1) for k in lay.keys("flows.0.")]).cuda().requires_grad_(True)
        loss = tf_fm_loss(lay, flat, x, t, z, cond, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
        loss.backward()
        runs.append((loss.detach().clone(), flat.grad.detach().clone()))
    _same(runs)


def test_wide_epic_gradients_are_bitwise_repeatable():
    from particle_fm_amd.fm_loss_wide import epic_wide_fm_loss
    from particle_fm_amd.layout_wide import EpicWideLayout
    from tests.conftest import load_wide_golden
    from tests.test_layout_cpu import cfg_of
    g = load_wide_golden("jetclass")  # BASELINE cfg 5 shapes: H = 300, 20 layers, N = 128
    lay = EpicWideLayout(cfg_of(g.hp))
    tag = "loss_f32/"
    x, t, z, mask, cond = _tile([g.get(tag + k).cuda() for k in ("x", "t", "z", "mask", "cond")], 16)
    runs = []
    for _ in range(REPS):
        state = {k[len("flows.0.net."):]: v.cuda().requires_grad_(True) for k, v in g.state.items() if k.startswith("flows.0.net.")}
        src = lay.source_vector(state, "", freqs=g.freqs)
        loss = epic_wide_fm_loss(lay, src, x, t, z, cond, mask, 1e-4, "FM-OT")
        loss.backward()
        runs.append((loss.detach().clone(), torch.cat([p.grad.reshape(-1) for p in state.values()]).clone()))
    _same(runs)


def test_cross_attention_gradients_are_bitwise_repeatable():
    from particle_fm_amd.fm_loss_ca import ca_fm_loss
    from particle_fm_amd.layout_ca import CaConfig, CaLayout
    from tests.conftest import load_ca_golden
    g = load_ca_golden("lhco")
    lay = CaLayout(CaConfig.from_hparams(g.hp))
    tag = "loss_f32/"
    x, t, mask, cond, z = _tile([g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond", "z")], 12)
    runs = []
    for _ in range(REPS):
        flat = torch.cat([g.state[k].reshape(-1) for k in lay.keys("flows.0.")]).cuda().requires_grad_(True)
        loss = ca_fm_loss(lay, flat, x, t, z, cond, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
        loss.backward()
        runs.append((loss.detach().clone(), flat.grad.detach().clone()))
    _same(runs)


def test_mdma_gradients_are_bitwise_repeatable():
    from particle_fm_amd.fm_loss_mdma import mdma_fm_loss
    from particle_fm_amd.layout_mdma import MdmaConfig, MdmaLayout
    from tests.conftest import load_mdma_golden
    g = load_mdma_golden("yaml")
    lay = MdmaLayout(MdmaConfig.from_hparams(g.hp))
    tag = "loss_f32/"
    x, t, mask, z = _tile([g.get(tag + k).cuda() for k in ("x", "t", "mask", "z")], 16)
    runs = []
    for _ in range(REPS):
        flat = torch.cat([g.state[k].reshape(-1) for k in lay.keys("flows.0.")]).cuda().requires_grad_(True)
        loss = mdma_fm_loss(lay, flat, x, t, z, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
        loss.backward()
        runs.append((loss.detach().clone(), flat.grad.detach().clone()))
    _same(runs)
