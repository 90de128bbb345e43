"""Host logic of the drop-in boundary (no GPU): reference state_dict layout, initialisation stream,
error behaviour, C ABI exports, flat-parameter aliasing."""
import ctypes
import re

import pytest
import torch

from particle_fm_amd import _lib
from particle_fm_amd.engine import FlatParams
from particle_fm_amd.models import CNF, SetFlowMatchingLitModule
from tests.conftest import ROOT, load_golden


def _yaml_kwargs(hp):
    keys = ["model", "features", "hidden_dim", "num_particles", "frequencies", "layers", "latent", "activation",
            "wrapper_func", "t_local_cat", "t_global_cat", "add_time_to_input", "t_emb", "loss_type",
            "global_cond_dim", "local_cond_dim", "dropout", "sum_scale"]
    return {k: hp[k] for k in keys}


def test_state_dict_keys_shapes_and_init_match_reference(golden):
    # oracle/make_golden.py builds the reference CNF under manual_seed(12345); same stream here
    torch.manual_seed(12345)
    cnf = CNF(**_yaml_kwargs(golden.hp))
    sd = {f"flows.0.{k}": v for k, v in cnf.state_dict().items()}
    assert list(sd.keys()) == golden.keys
    for k in golden.keys:
        assert sd[k].shape == golden.state[k].shape, k
        if k.endswith("weight_v") or k.endswith("frequencies"):
            # weight_v and the buffer are untouched by the fixture's perturbation: bit-identical init
            assert torch.equal(sd[k], golden.state[k]), k


def test_lit_module_surface():
    m = SetFlowMatchingLitModule(optimizer=None, features=3, hidden_dim=128, num_particles=30, frequencies=16,
                                 layers=2, latent=10, t_local_cat=True, t_global_cat=True, add_time_to_input=False,
                                 t_emb="cosine", sigma=1e-4)
    for name in ("num_particles", "features", "loss_type", "sigma", "use_normaliser", "optimizer", "scheduler"):
        assert name in m.hparams
    assert m.hparams.num_particles == 30
    g = load_golden("jetnet30")
    # strict load of a reference-shaped state_dict (checkpoint / EMA swap path, callbacks/ema.py:145-157)
    m6 = SetFlowMatchingLitModule(optimizer=None, **{k: v for k, v in _yaml_kwargs(g.hp).items()})
    # the reference registers the flows twice (self.flows and self.loss.flows, flow_matching_module.py:445-449),
    # so its checkpoints carry every tensor under "flows." and under "loss.flows."
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m6.load_state_dict(full, strict=True)
    assert list(m6.state_dict().keys()) == g.keys + ["loss." + k for k in g.keys]  # and nothing cached leaks in
    assert sum(p.numel() for p in m6.parameters()) == 561330
    for fn in ("training_step", "validation_step", "sample", "configure_optimizers", "forward"):
        assert callable(getattr(m6, fn))


@pytest.mark.parametrize("kw,exc", [
    (dict(model="transformer"), NotImplementedError),           # flow_matching_module.py:170
    (dict(loss_type="nope"), NotImplementedError),              # :465
    (dict(t_emb="nope"), NotImplementedError),                  # :231
    (dict(criterion="l1"), NotImplementedError),                # losses.py:36
    (dict(use_normaliser=True, normaliser_config={"extra_dims": (0,)}), NotImplementedError),
    (dict(loss_type="CFM-OT"), NotImplementedError),
    (dict(model="mdma"), ValueError),  # MDMA's defaults concatenate the time embedding with Linears sized by ITS frequencies (6), not the model's (16)
    (dict(model="droid_fulltransformer", net_config={"te_config": {"model_dim": 64, "mha_config": {"num_heads": 4}}}),
     NotImplementedError),
])
def test_constructor_errors(kw, exc):
    base = dict(optimizer=None, features=3, hidden_dim=128, num_particles=30, frequencies=16, layers=1, latent=10,
                t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="cosine")
    base.update(kw)
    with pytest.raises(exc):
        SetFlowMatchingLitModule(**base)


def test_no_cpu_fallback():
    m = SetFlowMatchingLitModule(optimizer=None, features=3, hidden_dim=128, num_particles=30, frequencies=16,
                                 layers=1, latent=10, t_local_cat=True, t_global_cat=True, add_time_to_input=False,
                                 t_emb="cosine")
    x = torch.randn(2, 30, 3)
    with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
        m.flows[0](torch.rand(2), x)
    with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
        m.flows[0].decode(x, None, None, ode_solver="rk4")  # has a HIP path, not a CPU one
    with pytest.raises(NotImplementedError):
        m.flows[0].decode(x, None, None, ode_solver="ieuler")
    with pytest.raises(NotImplementedError):
        m.flows[0].decode(x, None, None, ode_solver="bogus")
    with pytest.raises(SyntaxError):
        m.flows[0].decode(x, None, None, ode_solver="em")  # flow_matching_module.py:326
    with pytest.raises(ValueError):
        mc = CNF(features=3, hidden_dim=128, num_particles=30, frequencies=16, layers=1, latent=10, t_local_cat=True,
                 t_global_cat=True, add_time_to_input=False, t_emb="cosine", global_cond_dim=2)
        mc(torch.rand(2), x)  # cond missing: epic.py:313-317


def test_c_abi_exports_every_declared_symbol():
    lib = _lib.load()
    import glob
    header = "".join(open(h).read() for h in sorted(glob.glob(f"{ROOT}/include/*.h")))
    declared = set(re.findall(r"^(?:int64_t|int|const char \*)\s*\*?(pfm_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    from particle_fm_amd.layout import PFM_ABI_VERSION
    assert lib.pfm_abi_version() == PFM_ABI_VERSION == 3
    from particle_fm_amd.layout import EpicConfig, EpicDesc, EpicLayout
    lay = EpicLayout(EpicConfig(num_particles=150, features=3, latent=10, layers=6, frequencies=16, t_local_cat=True,
                                t_global_cat=True))
    assert lib.pfm_epic_lds_bytes(ctypes.byref(lay.desc)) <= 163840
    assert lay.desc_floats * 4 >= ctypes.sizeof(EpicDesc)
    # host-side validation errors come back as codes + text, no GPU needed
    bad = EpicLayout(EpicConfig(num_particles=400, features=3, latent=10, layers=1, frequencies=16, t_local_cat=True,
                                t_global_cat=True))
    rc = lib.pfm_epic_forward(ctypes.byref(bad.desc), None, None, None, None, None, None, 1, None)
    assert rc == 10002 and b"LDS" in lib.pfm_last_error()


def test_tf_desc_mirror_matches_the_c_struct(tmp_path):
    """sizeof / offsetof of pfm_tf_desc as gcc sees them vs the ctypes mirror; host-side validation of the tf entry points."""
    import subprocess
    from particle_fm_amd.layout_tf import TfConfig, TfDesc, TfLayer, TfLayout
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pfm_tf.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(pfm_tf_desc), offsetof(pfm_tf_desc, layer), sizeof(pfm_tf_layer), offsetof(pfm_tf_desc, final_norm),'
                   'offsetof(pfm_tf_desc, o2));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", f"{ROOT}/include", str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(TfDesc), TfDesc.layer.offset, ctypes.sizeof(TfLayer), TfDesc.final_norm.offset, TfDesc.o2.offset]
    lib = _lib.load()
    lay = TfLayout(TfConfig(num_particles=279, global_cond_dim=5))
    assert lib.pfm_tf_workspace_floats(ctypes.byref(lay.desc), 128, 0) > 0
    assert lib.pfm_tf_workspace_floats(ctypes.byref(lay.desc), 128, 1) > lib.pfm_tf_workspace_floats(ctypes.byref(lay.desc), 128, 0)
    lay.desc.head_dim = 32
    assert lib.pfm_tf_workspace_floats(ctypes.byref(lay.desc), 1, 0) == -1
    rc = lib.pfm_tf_forward(ctypes.byref(lay.desc), None, None, 0, None, None, None, None, 1, None, None)
    assert rc == 10001 and b"head_dim" in lib.pfm_last_error()


def test_ca_desc_mirror_matches_the_c_struct(tmp_path):
    """pfm_ca_desc as gcc lays it out vs the ctypes mirror; host-side validation of the cross-attention entry points."""
    import subprocess
    from particle_fm_amd.layout_ca import CaConfig, CaDesc, CaLayer, CaLayout
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pfm_ca.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(pfm_ca_desc), offsetof(pfm_ca_desc, from_layer), sizeof(pfm_ca_layer), offsetof(pfm_ca_desc, to_layer),'
                   'offsetof(pfm_ca_desc, global_tokens), offsetof(pfm_ca_desc, o2));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", f"{ROOT}/include", str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(CaDesc), CaDesc.from_layer.offset, ctypes.sizeof(CaLayer), CaDesc.to_layer.offset,
                   CaDesc.global_tokens.offset, CaDesc.o2.offset]
    lib = _lib.load()
    lay = CaLayout(CaConfig(num_particles=279, global_cond_dim=5))
    n0, n1 = (lib.pfm_ca_workspace_floats(ctypes.byref(lay.desc), 128, tr) for tr in (0, 1))
    assert 0 < n0 < n1
    assert lib.pfm_ca_backward_scratch_floats(ctypes.byref(lay.desc), 128) > 0
    lay.desc.tokens = 9
    assert lib.pfm_ca_workspace_floats(ctypes.byref(lay.desc), 1, 0) == -1
    rc = lib.pfm_ca_forward(ctypes.byref(lay.desc), None, None, 0, None, None, None, None, 1, None, None)
    assert rc == 10001 and b"tokens" in lib.pfm_last_error()


def test_mdma_desc_mirror_matches_the_c_struct(tmp_path):
    """pfm_mdma_desc as gcc lays it out vs the ctypes mirror; host-side validation of the MDMA entry points."""
    import subprocess
    from particle_fm_amd.layout_mdma import MdmaBlock, MdmaConfig, MdmaDesc, MdmaLayout
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pfm_mdma.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(pfm_mdma_desc), offsetof(pfm_mdma_desc, block), sizeof(pfm_mdma_block), offsetof(pfm_mdma_desc, c_cat),'
                   'offsetof(pfm_mdma_desc, out_b), offsetof(pfm_mdma_block, fc2c_b));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", f"{ROOT}/include", str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(MdmaDesc), MdmaDesc.block.offset, ctypes.sizeof(MdmaBlock), MdmaDesc.c_cat.offset,
                   MdmaDesc.out_b.offset, MdmaBlock.fc2c_b.offset]
    lib = _lib.load()
    lay = MdmaLayout(MdmaConfig(num_particles=150, hidden=128, num_layers=4, frequencies=16))
    n0, n1 = (lib.pfm_mdma_workspace_floats(ctypes.byref(lay.desc), 128, tr) for tr in (0, 1))
    assert 0 < n0 < n1
    assert lib.pfm_mdma_backward_scratch_floats(ctypes.byref(lay.desc), 128) > 0
    lay.desc.latent = 10
    assert lib.pfm_mdma_workspace_floats(ctypes.byref(lay.desc), 1, 0) == -1
    rc = lib.pfm_mdma_forward(ctypes.byref(lay.desc), None, None, 0, None, None, None, None, 1, None, None)
    assert rc == 10001 and b"latent" in lib.pfm_last_error()


def test_flat_params_alias_and_survive_load_state_dict():
    m = SetFlowMatchingLitModule(optimizer=None, features=3, hidden_dim=128, num_particles=30, frequencies=16,
                                 layers=1, latent=10, t_local_cat=True, t_global_cat=True, add_time_to_input=False,
                                 t_emb="cosine")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m.parameters())
    assert fp.is_intact()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    m.load_state_dict({k: v + 1 if v.is_floating_point() and "freq" not in k else v for k, v in before.items()})
    assert fp.is_intact()
    p0 = next(m.parameters())
    assert torch.equal(fp.flat[: p0.numel()].view_as(p0), p0.data)
    fp.grad.fill_(2.0)
    assert float(p0.grad.sum()) == 2.0 * p0.numel()


def test_normaliser_buffers_and_errors():
    """use_normaliser=True: IterativeNormLayer buffers appear under the reference's names; unsupported shapes / CPU inputs raise."""
    from particle_fm_amd.models.components.norm_layer import IterativeNormLayer
    m = SetFlowMatchingLitModule(optimizer=None, features=3, hidden_dim=128, num_particles=30, frequencies=16, layers=1, latent=10,
                                 t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="cosine", global_cond_dim=2,
                                 use_normaliser=True, normaliser_config={"max_n": 2000})
    keys = list(m.state_dict().keys())
    assert keys[-8:] == ["normaliser.means", "normaliser.vars", "normaliser.n", "normaliser.m2",
                         "ctxt_normaliser.means", "ctxt_normaliser.vars", "ctxt_normaliser.n", "ctxt_normaliser.m2"]
    assert m.normaliser.means.shape == (1, 3) and m.normaliser.n.dtype == torch.int64 and m.normaliser.max_n == 2000
    with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
        m.normaliser(torch.randn(2, 30, 3), torch.ones(2, 30, dtype=torch.bool))
    with pytest.raises(NotImplementedError):
        IterativeNormLayer((30, 3), extra_dims=(0,))
    with pytest.raises(ValueError):
        IterativeNormLayer((3,), means=torch.zeros(1, 3))


def test_epic_path_is_chosen_by_set_size_and_width():
    """configs/experiment/lhco/{x_jet,y_jet}.yaml (N = 279) and whole_event.yaml (N = 560) run flow_matching.yaml at hidden 128: the
    set does not fit the jet-resident kernel's LDS tile, the module must take the row-matrix path instead of raising."""
    from particle_fm_amd.layout import EpicLayout
    from particle_fm_amd.layout_wide import EpicWideLayout
    from particle_fm_amd.models import SetFlowMatchingLitModule
    from particle_fm_amd.models.components.epic import jet_resident_fits
    assert jet_resident_fits(150, 3) and not jet_resident_fits(152, 3) and not jet_resident_fits(279, 3)
    kw = dict(optimizer=None, model="epic", features=3, hidden_dim=128, frequencies=16, layers=2, latent=10, t_local_cat=True,
              t_global_cat=True, add_time_to_input=False, t_emb="cosine", loss_type="FM-OT")
    for n, gc in ((279, 4), (560, 4)):
        m = SetFlowMatchingLitModule(num_particles=n, global_cond_dim=gc, local_cond_dim=gc, **kw)
        net = m.flows[0].net
        assert net.wide and net.is_wide(n) and isinstance(net.layout(), EpicWideLayout)
        assert not net.is_wide(150) and isinstance(net.layout(150), EpicLayout)  # sample(num_points=150) would stay jet-resident
    m = SetFlowMatchingLitModule(num_particles=150, **kw)
    assert not m.flows[0].net.wide and isinstance(m.flows[0].net.layout(), EpicLayout)
    assert m.flows[0].net.is_wide(279)
    m = SetFlowMatchingLitModule(num_particles=30, **dict(kw, hidden_dim=300, latent=16))
    assert m.flows[0].net.wide


def test_freq_table_override():
    """The cosine embedding's frequency table is part of what a checkpoint means (freq_table.py): default = host-independent,
    "torch" = this host's fp32 exp, or the recorded table."""
    import numpy as np

    from particle_fm_amd.models import SetFlowMatchingLitModule
    kw = dict(optimizer=None, model="epic", features=3, hidden_dim=128, num_particles=30, frequencies=16, layers=1, latent=10,
              t_local_cat=True, t_global_cat=True, add_time_to_input=False, t_emb="cosine", loss_type="FM-OT")
    m = SetFlowMatchingLitModule(**kw)
    net = m.flows[0].net
    lay = net.layout()
    d = lay.desc
    f0 = net.packed_weights()[d.freqs:d.freqs + 32]
    assert torch.equal(f0, torch.arange(32, dtype=torch.float64).exp().float())
    m.set_freq_table("torch")
    assert torch.equal(net.packed_weights()[d.freqs:d.freqs + 32], torch.arange(32).exp())
    rec = torch.arange(32).exp()
    rec[15] = torch.from_numpy(np.nextafter(rec[15:16].numpy(), np.float32(np.inf)))[0]  # the 1-ulp difference measured between hosts
    m.set_freq_table(rec)
    got = net.packed_weights()[d.freqs:d.freqs + 32]
    assert torch.equal(got, rec) and not torch.equal(got, f0)
    m.set_freq_table("float64-rounded")
    assert torch.equal(net.packed_weights()[d.freqs:d.freqs + 32], f0)
    with pytest.raises(ValueError):
        m.set_freq_table(torch.ones(5))
    with pytest.raises(ValueError):
        m.set_freq_table("nope")
    assert "freq" not in " ".join(k for k in m.state_dict() if "frequencies" not in k)  # nothing new in state_dict
