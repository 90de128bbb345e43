"""layout_ca.py: descriptor, gather maps and blob formats, checked on the CPU against the reference vectors."""
import numpy as np
import pytest
import torch

from particle_fm_amd.layout_ca import CaConfig, CaLayout
from tests import ca_blob_interp


def _layout(g):
    return CaLayout(CaConfig.from_hparams(g.hp))


def test_state_dict_order_and_count(ca_golden):
    lay = _layout(ca_golden)
    assert lay.keys("flows.0.") == [k for k in ca_golden.keys if not k.endswith("frequencies")]
    assert lay.n_params == sum(v.numel() for k, v in ca_golden.state.items() if not k.endswith("frequencies"))


def test_blob_evaluates_to_reference(ca_golden):
    g = ca_golden
    lay = _layout(g)
    blob = lay.pack_blob(g.state, "flows.0.", g.freqs)
    assert blob.numel() == lay.blob_total
    tag = "nfe_f32/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = ca_blob_interp.forward(lay.desc, blob, t, x, cond, mask.reshape(x.shape[0], -1).float())
    torch.testing.assert_close(v, g.get(tag + "v_vec_t"), rtol=2e-4, atol=2e-5)


def test_grad_pos_is_a_bijection_onto_primary_slots(ca_golden):
    lay = _layout(ca_golden)
    gp = lay.grad_pos
    assert gp.shape == (lay.n_params,) and len(np.unique(gp)) == lay.n_params
    assert np.array_equal(lay.index_map[gp], np.arange(lay.n_params))


def test_unsupported_configs_are_rejected():
    with pytest.raises(NotImplementedError):
        CaLayout(CaConfig(num_particles=30, model_dim=64, num_heads=4, hidden=128))
    with pytest.raises(NotImplementedError):
        CaLayout(CaConfig(num_particles=30, model_dim=128, num_heads=4))   # head_dim 32
    with pytest.raises(NotImplementedError):
        CaLayout(CaConfig(num_particles=30, model_dim=128, num_heads=8, num_tokens=8))  # 8 tokens x head_dim 16 > 64
