"""Host logic: the weight blob (particle_fm_amd/layout.py) evaluated by a CPU interpreter must
reproduce the oracle / the reference's golden outputs; descriptor sanity; gradient flow."""
import ctypes

import numpy as np
import pytest
import torch

from oracle.fm_ref import EpicVectorField
from particle_fm_amd.layout import EpicConfig, EpicDesc, EpicLayout
from tests.blob_interp import interp_forward


def cfg_of(hp):
    return EpicConfig(
        num_particles=hp["num_particles"], features=hp["features"], hidden_dim=hp["hidden_dim"],
        latent=hp["latent"], layers=hp["layers"], frequencies=hp["frequencies"],
        t_local_cat=hp["t_local_cat"], t_global_cat=hp["t_global_cat"],
        global_cond_dim=hp["global_cond_dim"], local_cond_dim=hp["local_cond_dim"], sum_scale=hp["sum_scale"],
        t_emb=hp.get("t_emb", "cosine"), add_time_to_input=bool(hp.get("add_time_to_input", False)),
        neg_slope={"leaky_relu": 0.01, "relu": 0.0}.get(hp.get("activation", "leaky_relu"), 1.0),  # (epic.py:180; components/epic.py::activation_slope)
    )


def test_param_count_matches_reference():
    cfg = EpicConfig(num_particles=150, features=3, hidden_dim=128, latent=10, layers=6, frequencies=16,
                     t_local_cat=True, t_global_cat=True)
    assert cfg.param_count() == 561330  # SURVEY.md §8, BASELINE.md §2


@pytest.mark.parametrize("mk", ["f32", "none"])
def test_blob_interpreter_matches_reference_vectors(golden, mk):
    lay = EpicLayout(cfg_of(golden.hp))
    blob = lay.pack_blob(golden.state, "flows.0.net.", freqs=golden.freqs)
    assert blob.shape == (lay.blob_total,)
    assert bytes(lay.desc) == blob[lay.desc.blob_floats:].numpy().tobytes()[: len(bytes(lay.desc))]
    tag = f"nfe_{mk}/"
    x, t = golden.get(tag + "x"), golden.get(tag + "t")
    v = interp_forward(lay, blob, t, x, golden.get(tag + "cond"), golden.get(tag + "mask"))
    torch.testing.assert_close(v, golden.get(tag + "v_vec_t"), atol=1e-5, rtol=1e-4)


def test_offsets_aligned_and_disjoint(golden):
    lay = EpicLayout(cfg_of(golden.hp))
    d = lay.desc
    offs = [d.freqs, d.l1x.W, d.l1_We, d.l1_b, d.l2.A, d.l2.AT, d.l2.We, d.l2.b, d.g1.W, d.g1.b, d.g2.W, d.g2.b,
            d.l3_W, d.l3_We, d.l3_b]
    for k in range(d.layers):
        ly = d.layer[k]
        offs += [ly.gl1.W, ly.gl1.b, ly.gl2.W, ly.gl2.b, ly.lc1.A, ly.lc1.AT, ly.lc1.We, ly.lc1.b,
                 ly.lc2.A, ly.lc2.AT, ly.lc2.We, ly.lc2.b]
    assert all(o % 4 == 0 and 0 <= o < d.blob_floats for o in offs)
    assert len(set(offs)) == len(offs)
    assert lay.index_map.min() >= 0 and lay.index_map.max() < lay.n_source
    assert ctypes.sizeof(EpicDesc) < 4096  # travels as a kernel argument


def test_t_cat_off_maps_to_zero_columns():
    cfg = EpicConfig(num_particles=20, features=3, latent=8, layers=1, frequencies=4, t_local_cat=False, t_global_cat=True)
    lay = EpicLayout(cfg)
    d = lay.desc
    # extras of the local linears must point at the zero slot for the T time columns
    We = lay.index_map[d.l2.We: d.l2.We + cfg.t_dim * 128]
    assert np.all(We == lay.zero_off)


def test_blob_gradient_reaches_weight_norm_params(golden):
    lay = EpicLayout(cfg_of(golden.hp))
    state = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in golden.state.items()}
    blob = lay.pack_blob(state, "flows.0.net.", freqs=golden.freqs)
    tag = "nfe_f32/"
    v = interp_forward(lay, blob, golden.get(tag + "t"), golden.get(tag + "x"), golden.get(tag + "cond"), golden.get(tag + "mask"))
    v.square().sum().backward()
    # reference graph through the oracle
    st2 = {k: v_.clone().requires_grad_(v_.is_floating_point() and "frequencies" not in k) for k, v_ in golden.state.items()}
    vf = EpicVectorField(st2, "flows.0.net", golden.hp, freqs=golden.freqs)
    x, t = golden.get(tag + "x"), golden.get(tag + "t")
    v2 = vf(t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1), x, cond=golden.get(tag + "cond"), mask=golden.get(tag + "mask"))
    v2.square().sum().backward()
    for k in st2:
        if st2[k].grad is None:
            continue
        scale = max(st2[k].grad.abs().max().item(), 1e-8)
        assert (state[k].grad - st2[k].grad).abs().max().item() <= 5e-5 * scale + 1e-7, k


def test_kq16_wq16_copies_hold_the_same_weights_as_the_km16_blocks(golden):
    """The lean sampler's chains read second copies of the per-jet GEMV blocks (KQ16 / WQ16, include/pfm_hip.h): element for element
    the weights the KM16 / KP16 blocks carry behind their time (and conditioning) rows; forward-only (no gradient slot points there)."""
    lay = EpicLayout(cfg_of(golden.hp))
    blob = lay.pack_blob(golden.state, "flows.0.net.", freqs=golden.freqs).numpy()
    d, cfg = lay.desc, lay.cfg
    T, C, H, L = cfg.t_dim, cfg.global_cond_dim, 128, cfg.latent
    km16 = lambda off, k, o: blob[off + ((k >> 4) * 32 + (o >> 2)) * 64 + (k & 15) * 4 + (o & 3)]
    kq16 = lambda off, k, o: blob[off + (k >> 4) * 2048 + o * 16 + (k & 15)]
    kp16 = lambda off, k, o: blob[off + k * 16 + o]
    wq16 = lambda off, k, o: blob[off + (k >> 4) * 256 + o * 16 + (k & 15)]
    k272, o128 = np.arange(2 * H + 16)[:, None], np.arange(H)[None, :]
    k128, o16 = np.arange(H)[:, None], np.arange(16)[None, :]
    assert np.array_equal(kq16(d.q_g1, k272[: 2 * H], o128), km16(d.g1.W, T + C + k272[: 2 * H], o128))
    assert np.array_equal(wq16(d.q_g2, k128, o16), kp16(d.g2.W, T + C + k128, o16))
    for layer in range(d.layers):
        ly = d.layer[layer]
        assert np.array_equal(kq16(d.q_gl1[layer], k272[: 2 * H + L], o128), km16(ly.gl1.W, T + C + k272[: 2 * H + L], o128))
        assert np.array_equal(wq16(d.q_gl2[layer], k128, o16), kp16(ly.gl2.W, T + C + k128, o16))
        kg = np.arange(L)[:, None]
        assert np.array_equal(kq16(d.q_we1[layer], kg, o128), km16(ly.lc1.We, T + cfg.local_cond_dim + kg, o128))
        assert np.any(kq16(d.q_gl1[layer], k272, o128) != 0) and np.all(kq16(d.q_gl1[layer], k272[2 * H + L:], o128) == 0)  # g padded to 16 rows
    for off in (d.q_g1, d.q_g2, d.q_gl1[0], d.q_gl2[0], d.q_we1[0]):
        assert off % 4 == 0 and np.all(lay.grad_index_map[off: off + 2048] == lay.zero_off)
