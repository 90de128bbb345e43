#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own hot-path modules.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Runs only in the build
container, where ``/root/reference`` is mounted; the GPU box never sees the
reference, only the committed ``.npz`` vectors this script writes.  Nothing
from the reference is copied: its files are imported *where they lie* and only
numeric inputs/outputs are stored.

Loader recipe: the reference package's ``__init__``
pulls plotting / jetnet / lightning, none of which is installed, so the five
hot-path files are loaded by path behind empty parent packages, with inert
stand-ins for the third-party names they import but the path never calls
(``ot``, ``pytorch_lightning.LightningModule``, ``torchdyn.core.NeuralODE``,
``zuko.utils.odeint``).  Consequence: ``CNF.decode`` (torchdyn) cannot run, so
the "midpoint" vectors are *reference vector field + restated integrator*
(``oracle/fm_ref.py::midpoint_trajectory_end``).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import logging
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PFM_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

from oracle.fm_ref import midpoint_trajectory_end  # noqa: E402


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def load_reference():
    """Import the reference hot-path modules by file path.  Returns a namespace."""
    for name in ("particle_fm", "particle_fm.utils", "particle_fm.models", "particle_fm.models.components"):
        _pkg(name)
    pl = types.ModuleType("particle_fm.utils.pylogger")
    pl.get_pylogger = logging.getLogger
    sys.modules["particle_fm.utils.pylogger"] = pl
    sys.modules["ot"] = types.ModuleType("ot")
    plm = types.ModuleType("pytorch_lightning")
    plm.LightningModule = torch.nn.Module
    sys.modules["pytorch_lightning"] = plm
    td, tdc = _pkg("torchdyn"), types.ModuleType("torchdyn.core")
    tdc.NeuralODE = None
    sys.modules["torchdyn.core"] = tdc
    zk, zku = _pkg("zuko"), types.ModuleType("zuko.utils")
    zku.odeint = None
    sys.modules["zuko.utils"] = zku

    def load(mod, rel):
        spec = importlib.util.spec_from_file_location(mod, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[mod] = m
        spec.loader.exec_module(m)
        return m

    comp = sys.modules["particle_fm.models.components"]
    base = "particle_fm/models/components/"
    mods = {}
    for name in ("diffusion", "epic", "time_emb", "droid_transformer", "norm_layer", "solver", "mdma", "losses", "mlp"):
        mods[name] = load(f"particle_fm.models.components.{name}", base + name + ".py")
    comp.EPiC_encoder = mods["epic"].EPiC_encoder
    comp.MDMA = mods["mdma"].MDMA
    comp.IterativeNormLayer = mods["norm_layer"].IterativeNormLayer
    fmm = load("particle_fm.models.flow_matching_module", "particle_fm/models/flow_matching_module.py")
    ns = types.SimpleNamespace(**mods)
    ns.fmm = fmm
    return ns


# ----------------------------------------------------------------------------------------------
# configurations (reference yaml: configs/model/flow_matching.yaml + experiment overrides)
# ----------------------------------------------------------------------------------------------
BASE = dict(
    model="epic", features=3, hidden_dim=128, frequencies=16, layers=6, latent=10,
    activation="leaky_relu", wrapper_func="weight_norm", t_local_cat=True, t_global_cat=True,
    add_time_to_input=False, t_emb="cosine", loss_type="FM-OT", global_cond_dim=0, local_cond_dim=0,
    dropout=0.0, sum_scale=1e-2,
)
CONFIGS = {
    # BASELINE cfg 2 (experiment/jetnet/fm_tops30.yaml) and cfg 3 (fm_tops150.yaml): same net, N differs
    "jetnet30": dict(BASE, num_particles=30),
    "jetnet150": dict(BASE, num_particles=150),
    # conditioned variants of the same operator at reduced depth (keeps the fixture small):
    # fm_tops150_cond.yaml style (global 2 / local 2) and jetclass_cond.yaml style (global 12 / local 0, F=13, L=16)
    "cond_gl": dict(BASE, num_particles=40, layers=2, global_cond_dim=2, local_cond_dim=2),
    "cond_jetclass": dict(BASE, num_particles=48, layers=2, features=13, latent=16, global_cond_dim=12),
    # t_emb="gaussian" (flow_matching_module.py:178-181, 213-221; time_emb.py:9-22): a learned time embedding -- random Fourier
    # features -> Linear -> activation -> Linear(2 * frequencies) -- whose parameters train with the network
    "gauss": dict(BASE, num_particles=24, layers=2, t_emb="gaussian", global_cond_dim=2, local_cond_dim=2),
    # activation (epic.py:180: getattr(F, activation, lambda x: x)): "relu", and a name torch.nn.functional does not have = none
    "relu": dict(BASE, num_particles=24, layers=2, activation="relu", global_cond_dim=2),
    "noact": dict(BASE, num_particles=24, layers=1, activation="none"),
    # add_time_to_input=True for model "epic" (flow_matching_module.py:126, 199-200: the network sees cat(time embedding, x); the class
    # default, off in configs/model/flow_matching.yaml): with t_local_cat (fc_l1 then has TWO time blocks) and without
    "addtime": dict(BASE, num_particles=24, layers=2, add_time_to_input=True, global_cond_dim=2, local_cond_dim=2),
    "addtime_notl": dict(BASE, num_particles=20, layers=2, add_time_to_input=True, t_local_cat=False),
}


def make_mask(B, N, kind, gen):
    if kind == "none":
        return None
    n = torch.randint(max(2, N // 5), N + 1, (B,), generator=gen)
    n[0] = N  # one full jet
    if B > 1:
        n[1] = max(2, N // 5)  # one with a long padded tail
    m = (torch.arange(N)[None, :] < n[:, None]).unsqueeze(-1)
    return m.to(torch.int64) if kind == "int64" else m.to(torch.float32)


def gen_config(ref, name, hp, out_dir, B=4, seed=12345):
    torch.manual_seed(seed)
    cnf = ref.fmm.CNF(**hp)
    # default init leaves weight_g = ||v|| (identity reparam); perturb g and bias so g matters
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in cnf.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif k.endswith("bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=gen))
    flows = torch.nn.ModuleList([cnf])
    state = {f"flows.0.{k}": v.detach().clone() for k, v in cnf.state_dict().items()}
    N, Fe, Cg = hp["num_particles"], hp["features"], hp["global_cond_dim"]
    out = {"_keys": np.array(list(state.keys()))}
    for k, v in state.items():
        out["sd/" + k] = v.numpy()
    out["hp_json"] = np.array(__import__("json").dumps(hp))
    # the table time_emb.py:90 evaluates on THIS machine (fp32 exp is host-dependent, see oracle/fm_ref.py)
    out["freqs"] = torch.arange(2 * hp["frequencies"]).exp().numpy()

    for mk in ("f32", "int64", "none"):
        mask = make_mask(B, N, mk, gen)
        x = torch.randn(B, N, Fe, generator=gen)
        if mask is not None:
            x = x * mask
        cond = torch.randn(B, Cg, generator=gen) if Cg > 0 else None
        t = torch.rand(B, generator=gen)
        tag = f"nfe_{mk}/"
        with torch.no_grad():
            # training-style call: t is (B,N)
            tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
            temb = cnf.time_embedding(tt, x, hp["t_emb"])
            v_vec = cnf(tt, x, cond=cond, mask=mask)
            # sampling-style call: 0-dim t (same value for every jet)
            ts = t[0].clone()
            v_sca = cnf(ts, x, cond=cond, mask=mask)
        out[tag + "x"] = x.numpy()
        out[tag + "t"] = t.numpy()
        out[tag + "temb"] = temb[:, 0, :].numpy()
        if mask is not None:
            out[tag + "mask"] = mask.numpy()
        if cond is not None:
            out[tag + "cond"] = cond.numpy()
        out[tag + "v_vec_t"] = v_vec.numpy()
        out[tag + "v_scalar_t"] = v_sca.numpy()

    # ---- loss + grads through the reference's own FlowMatchingLoss (draws replayed by seed) ----
    for mk in ("f32", "none"):
        mask = make_mask(B, N, mk, gen)
        x = torch.randn(B, N, Fe, generator=gen)
        if mask is not None:
            x = x * mask
        cond = torch.randn(B, Cg, generator=gen) if Cg > 0 else None
        loss_mod = ref.losses.FlowMatchingLoss(flows=flows, sigma=1e-4)
        s = 9999
        torch.manual_seed(s)
        cnf.zero_grad()
        loss = loss_mod(x, mask=mask, cond=cond)
        loss.backward()
        torch.manual_seed(s)  # replay the same draws, same order (losses.py:46, 53)
        t = torch.rand_like(torch.ones(B))
        z = torch.randn_like(x)
        tag = f"loss_{mk}/"
        out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
        if mask is not None:
            out[tag + "mask"] = mask.numpy()
        if cond is not None:
            out[tag + "cond"] = cond.numpy()
        out[tag + "loss"] = loss.detach().numpy()
        for k, p in cnf.named_parameters():
            if p.grad is not None:  # (GaussianFourierProjection.W is a frozen parameter)
                out[tag + "grad/flows.0." + k] = p.grad.detach().clone().numpy()

    # ---- CFM loss (losses.py:101-136), mask required ----
    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen) if Cg > 0 else None
    loss_mod = ref.losses.ConditionalFlowMatchingLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(4242)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    torch.manual_seed(4242)  # losses.py:104, 108, 116
    t = torch.rand_like(torch.ones(B))
    x0 = torch.randn_like(x)
    eps = torch.randn_like(x)
    tag = "cfm/"
    out[tag + "x"], out[tag + "t"], out[tag + "x0"], out[tag + "eps"] = x.numpy(), t.numpy(), x0.numpy(), eps.numpy()
    out[tag + "mask"] = mask.numpy()
    if cond is not None:
        out[tag + "cond"] = cond.numpy()
    out[tag + "loss"] = loss.detach().numpy()

    # ---- midpoint: reference vector field (imported CNF.forward via the imported ode_wrapper)
    #      + restated integrator ----
    for steps in (3, 10, 100):
        mask = make_mask(B, N, "f32", gen)
        cond = torch.randn(B, Cg, generator=gen) if Cg > 0 else None
        z = torch.randn(B, N, Fe, generator=gen)
        z0 = z * mask
        wrapped = ref.fmm.ode_wrapper(model=cnf, cond=cond, mask=mask, loss_type="FM-OT")
        with torch.no_grad():
            xe = midpoint_trajectory_end(wrapped, z0, torch.linspace(1.0, 0.0, steps))
        tag = f"midpoint_{steps}/"
        out[tag + "z"], out[tag + "mask"], out[tag + "x_end"] = z.numpy(), mask.numpy(), xe.numpy()
        if cond is not None:
            out[tag + "cond"] = cond.numpy()

    path = os.path.join(out_dir, f"epic_{name}.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


def gen_no_sets(ref, out_dir, seed=12345):
    """BASELINE cfg 1: fully-connected FM vector field (flow_matching_no_sets.py:41-66 + mlp.py:24-68).
    flow_matching_no_sets.py itself imports torchdyn at module top but only CNF.decode uses it."""
    spec = importlib.util.spec_from_file_location(
        "particle_fm.models.flow_matching_no_sets", os.path.join(REF, "particle_fm/models/flow_matching_no_sets.py")
    )
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    torch.manual_seed(seed)
    cnf = m.CNF(features=2, freqs=3)
    gen = torch.Generator().manual_seed(seed + 7)
    B = 16
    x = torch.randn(B, 2, generator=gen)
    cond = torch.zeros(B, 1)
    t = torch.rand(B, generator=gen)
    with torch.no_grad():
        v = cnf(t, x, cond=cond)
    flows = torch.nn.ModuleList([cnf])
    loss_mod = ref.losses.FlowMatchingLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(77)
    loss = loss_mod(x, mask=None, cond=cond)
    loss.backward()
    torch.manual_seed(77)  # losses.py:49, 53 (non-set branch: t per sample)
    tl = torch.rand_like(x[..., 0]).unsqueeze(-1)
    zl = torch.randn_like(x)
    out = {"_keys": np.array(list(cnf.state_dict().keys()))}
    for k, v_ in cnf.state_dict().items():
        out["sd/" + k] = v_.detach().numpy()
    out.update(x=x.numpy(), cond=cond.numpy(), t=t.numpy(), v=v.numpy(), loss_t=tl.numpy(), loss_z=zl.numpy(),
               loss=loss.detach().numpy())
    for k, p in cnf.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    # midpoint with restated integrator
    z = torch.randn(B, 2, generator=gen)
    wrapped = m.ode_wrapper(cnf, mask=None, cond=cond)
    with torch.no_grad():
        xe = midpoint_trajectory_end(wrapped, z, torch.linspace(1.0, 0.0, 20))
    out.update(mid_z=z.numpy(), mid_x_end=xe.numpy())
    path = os.path.join(out_dir, "no_sets_moons.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB")


# ----------------------------------------------------------------------------------------------
# BASELINE cfg 4: Full-Transformer vector field (configs/model/fm_droid_transformer.yaml:15-44)
# ----------------------------------------------------------------------------------------------
def tf_net_config(model_dim, num_layers, num_heads, ctxt_out=64):
    return dict(
        node_embd_config=dict(act_h="lrlu", nrm="layer"),
        ctxt_embd_config=dict(outp_dim=ctxt_out, act_h="lrlu", nrm="layer"),
        te_config=dict(model_dim=model_dim, num_layers=num_layers,
                       mha_config=dict(num_heads=num_heads, init_zeros=True, do_layer_norm=True),
                       dense_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True)),
        outp_embd_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True),
    )


TF_BASE = dict(model="droid_fulltransformer", features=3, frequencies=16, add_time_to_input=True, t_emb="cosine",
               loss_type="FM-OT")
TF_CONFIGS = {
    # reduced width/depth: every array (weights, all parameter gradients) is stored
    "small": (dict(TF_BASE, num_particles=40, global_cond_dim=3, net_config=tf_net_config(128, 2, 8)), 4, True),
    # the yaml's own sizes (experiment/lhco/jets_transformer.yaml:26-31): weights are re-derived from the seed
    # by oracle/seeded.py instead of being stored; gradients are stored sub-sampled
    "lhco": (dict(TF_BASE, num_particles=279, global_cond_dim=5, net_config=tf_net_config(256, 3, 16)), 2, False),
    "sincos": (dict(TF_BASE, num_particles=20, global_cond_dim=2, t_emb="sincos", frequencies=6,
                    net_config=tf_net_config(128, 1, 8)), 3, False),
    # t_emb="gaussian" (flow_matching_module.py:178-181, 213-221): the CNF's trainable embedding network in front of the field
    "gauss": (dict(TF_BASE, num_particles=24, global_cond_dim=2, t_emb="gaussian", hidden_dim=64, net_config=tf_net_config(128, 1, 8)), 3, False),
    # no conditioning, a smooth time embedding: the network CNF.log_prob can evaluate (flow_matching_module.py:337: self(t, x)) and an
    # adaptive solver converges on
    "plain": (dict(TF_BASE, num_particles=20, global_cond_dim=0, t_emb="sincos", frequencies=6, net_config=tf_net_config(128, 1, 8)), 3, False),
}


def gen_transformer(ref, name, hp, B, store_all, out_dir, seed=2024, file_prefix="tf"):
    import copy
    import json

    from oracle.seeded import seeded_state, subsample

    torch.manual_seed(seed)
    cnf = ref.fmm.CNF(**copy.deepcopy(hp))
    # init_zeros / output_init_zeros make the untrained field identically 0: every tensor is
    # replaced by a seeded, machine-independent draw (numpy PCG64) with the usual fan-in scale
    shapes = {k: tuple(v.shape) for k, v in cnf.state_dict().items() if k != "frequencies"}
    # default initialisation under the seed (incl. init_zeros / output_init_zeros): per-tensor sum and |.| sum
    init = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for k, v in cnf.state_dict().items()
                     if k != "frequencies"])
    new = seeded_state(shapes, seed)
    sd = cnf.state_dict()
    for k, v in new.items():
        sd[k] = torch.from_numpy(v)
    cnf.load_state_dict(sd)
    flows = torch.nn.ModuleList([cnf])
    N, Fe, Cg = hp["num_particles"], hp["features"], hp["global_cond_dim"]
    out = {"_keys": np.array(["flows.0." + k for k in cnf.state_dict().keys()])}
    out["_shapes_json"] = np.array(json.dumps({"flows.0." + k: list(s) for k, s in shapes.items()}))
    out["seed"] = np.array(seed)
    out["hp_json"] = np.array(json.dumps(hp))
    # cosine: the exp(arange) table of THIS machine (host-dependent, see oracle/fm_ref.py); sincos: the module buffer (:172)
    out["freqs"] = (torch.arange(2 * hp["frequencies"]).exp() if hp["t_emb"] == "cosine" else cnf.frequencies.clone()).numpy()
    out["abs_sum"] = np.array(sum(float(np.abs(v).sum(dtype=np.float64)) for v in new.values()))
    out["init_sums"] = init
    if store_all:
        for k, v in cnf.state_dict().items():
            out["sd/flows.0." + k] = v.detach().numpy()
    gen = torch.Generator().manual_seed(seed + 1)

    for mk in ("f32", "int64", "ones"):
        mask = make_mask(B, N, "f32" if mk == "ones" else mk, gen)
        if mk == "ones":
            mask = torch.ones_like(mask)
        x = torch.randn(B, N, Fe, generator=gen) * mask
        cond = torch.randn(B, Cg, generator=gen)
        t = torch.rand(B, generator=gen)
        tag = f"nfe_{mk}/"
        with torch.no_grad():
            tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
            temb = cnf.time_embedding(tt, x, hp["t_emb"])
            v_vec = cnf(tt, x, cond=cond, mask=mask)
            v_sca = cnf(t[0].clone(), x, cond=cond, mask=mask)
        out[tag + "x"], out[tag + "t"], out[tag + "temb"] = x.numpy(), t.numpy(), temb[:, 0, :].numpy()
        out[tag + "mask"], out[tag + "cond"] = mask.numpy(), cond.numpy()
        out[tag + "v_vec_t"], out[tag + "v_scalar_t"] = v_vec.numpy(), v_sca.numpy()

    def put_grads(tag):
        for k, p in cnf.named_parameters():
            if p.grad is None:  # MDMA: Block.cond_cls is constructed but never used (mdma.py:30, 36)
                continue
            g = p.grad.detach().clone().numpy()
            out[tag + "grad/flows.0." + k] = g if store_all else subsample(g)

    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen)
    loss_mod = ref.losses.FlowMatchingLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(9999)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    loss.backward()
    torch.manual_seed(9999)  # losses.py:46, 53
    t = torch.rand_like(torch.ones(B))
    z = torch.randn_like(x)
    tag = "loss_f32/"
    out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
    out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
    put_grads(tag)

    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen)
    loss_mod = ref.losses.ConditionalFlowMatchingLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(4242)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    loss.backward()
    torch.manual_seed(4242)  # losses.py:104, 108, 116
    t = torch.rand_like(torch.ones(B))
    x0 = torch.randn_like(x)
    eps = torch.randn_like(x)
    tag = "cfm/"
    out[tag + "x"], out[tag + "t"], out[tag + "x0"], out[tag + "eps"] = x.numpy(), t.numpy(), x0.numpy(), eps.numpy()
    out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
    put_grads(tag)

    for steps in ((3, 10, 100) if store_all else (3, 10)):
        mask = make_mask(B, N, "f32", gen)
        cond = torch.randn(B, Cg, generator=gen)
        z = torch.randn(B, N, Fe, generator=gen)
        wrapped = ref.fmm.ode_wrapper(model=cnf, cond=cond, mask=mask, loss_type="FM-OT")
        with torch.no_grad():
            xe = midpoint_trajectory_end(wrapped, z * mask, torch.linspace(1.0, 0.0, steps))
        tag = f"midpoint_{steps}/"
        out[tag + "z"], out[tag + "mask"], out[tag + "cond"], out[tag + "x_end"] = (
            z.numpy(), mask.numpy(), cond.numpy(), xe.numpy())


    # ---- DroidLoss (losses.py:304-342): y = x + t z, u = z mask ----
    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen)
    loss_mod = ref.losses.DroidLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(777)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    loss.backward()
    torch.manual_seed(777)  # losses.py:330, 335
    t = torch.rand_like(torch.ones(B))
    z = torch.randn_like(x)
    tag = "droid/"
    out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
    out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
    for k, p in [kp for kp in cnf.named_parameters() if kp[1].grad is not None][:6]:
        out[tag + "grad/flows.0." + k] = subsample(p.grad.detach().clone().numpy())
    path = os.path.join(out_dir, f"{file_prefix}_{name}.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# BASELINE cfg 5: EPiC at JetClass width (experiment/jetclass_cond.yaml:32-42): weights re-derived from the seed
# ----------------------------------------------------------------------------------------------
WIDE_BASE = dict(BASE, features=13, hidden_dim=300, latent=16, global_cond_dim=12, local_cond_dim=0)
WIDE_CONFIGS = {
    "small": (dict(WIDE_BASE, num_particles=24, layers=2), 4),
    "jetclass": (dict(WIDE_BASE, num_particles=128, layers=20), 2),
    # the class default t_emb="sincos" (flow_matching_module.py:104, 208-211) at hidden 128: both EPiC kernels
    "sincos": (dict(BASE, num_particles=24, layers=2, global_cond_dim=2, local_cond_dim=2, t_emb="sincos", frequencies=6), 4),
    # configs/experiment/lhco/x_jet.yaml:26-29 (y_jet.yaml the same): flow_matching.yaml at its default width with 279 particles and
    # 4 + 4 conditioning values -- hidden 128, but the set no longer fits the jet-resident kernel's LDS tile: row-matrix path
    "lhco128": (dict(BASE, num_particles=279, global_cond_dim=4, local_cond_dim=4), 2),
    # t_emb="gaussian" (flow_matching_module.py:178-181, 213-221) on the row-matrix path: the trainable embedding network in front
    "gauss": (dict(WIDE_BASE, num_particles=24, layers=2, t_emb="gaussian", local_cond_dim=12), 3),
    # no conditioning, a smooth time embedding (CNF.log_prob, see TF_CONFIGS["plain"]): hidden 128 = the jet-resident kernels, 136 = the row-matrix ones
    "plain": (dict(BASE, num_particles=24, layers=2, t_emb="sincos", frequencies=6), 3),
    "plainw": (dict(BASE, num_particles=24, layers=1, hidden_dim=136, latent=12, t_emb="sincos", frequencies=6), 3),
}


def gen_epic_wide(ref, name, hp, B, out_dir, seed=777):
    import json

    from oracle.seeded import seeded_state, subsample

    torch.manual_seed(seed)
    cnf = ref.fmm.CNF(**hp)
    shapes = {k: tuple(v.shape) for k, v in cnf.state_dict().items() if k != "frequencies"}
    new = seeded_state(shapes, seed)
    sd = cnf.state_dict()
    for k, v in new.items():
        sd[k] = torch.from_numpy(v)
    cnf.load_state_dict(sd)
    flows = torch.nn.ModuleList([cnf])
    N, Fe, Cg = hp["num_particles"], hp["features"], hp["global_cond_dim"]
    out = {"_keys": np.array(["flows.0." + k for k in cnf.state_dict().keys()])}
    out["_shapes_json"] = np.array(json.dumps({"flows.0." + k: list(s) for k, s in shapes.items()}))
    out["seed"] = np.array(seed)
    out["hp_json"] = np.array(json.dumps(hp))
    # cosine: the exp(arange) table of THIS machine (host-dependent, see oracle/fm_ref.py); sincos: the module buffer (:172)
    out["freqs"] = (torch.arange(2 * hp["frequencies"]).exp() if hp["t_emb"] == "cosine" else cnf.frequencies.clone()).numpy()
    out["abs_sum"] = np.array(sum(float(np.abs(v).sum(dtype=np.float64)) for v in new.values()))
    gen = torch.Generator().manual_seed(seed + 1)
    for mk in ("f32", "int64", "none"):
        mask = make_mask(B, N, mk, gen)
        x = torch.randn(B, N, Fe, generator=gen)
        if mask is not None:
            x = x * mask
        cond = torch.randn(B, Cg, generator=gen)
        t = torch.rand(B, generator=gen)
        tag = f"nfe_{mk}/"
        with torch.no_grad():
            tt = t.unsqueeze(-1).repeat_interleave(N, dim=1)
            v_vec = cnf(tt, x, cond=cond, mask=mask)
            v_sca = cnf(t[0].clone(), x, cond=cond, mask=mask)
        out[tag + "x"], out[tag + "t"], out[tag + "cond"] = x.numpy(), t.numpy(), cond.numpy()
        if mask is not None:
            out[tag + "mask"] = mask.numpy()
        out[tag + "v_vec_t"], out[tag + "v_scalar_t"] = v_vec.numpy(), v_sca.numpy()
    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen)
    loss_mod = ref.losses.FlowMatchingLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(9999)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    loss.backward()
    torch.manual_seed(9999)
    t = torch.rand_like(torch.ones(B))
    z = torch.randn_like(x)
    tag = "loss_f32/"
    out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
    out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
    for k, p in cnf.named_parameters():
        if p.grad is not None:  # (the random Fourier frequencies of t_emb="gaussian" are a frozen parameter, time_emb.py:18)
            out[tag + "grad/flows.0." + k] = subsample(p.grad.detach().clone().numpy())
    for steps in (3, 10):
        mask = make_mask(B, N, "f32", gen)
        cond = torch.randn(B, Cg, generator=gen)
        z = torch.randn(B, N, Fe, generator=gen)
        wrapped = ref.fmm.ode_wrapper(model=cnf, cond=cond, mask=mask, loss_type="FM-OT")
        with torch.no_grad():
            xe = midpoint_trajectory_end(wrapped, z * mask, torch.linspace(1.0, 0.0, steps))
        tag = f"midpoint_{steps}/"
        out[tag + "z"], out[tag + "mask"], out[tag + "cond"], out[tag + "x_end"] = (
            z.numpy(), mask.numpy(), cond.numpy(), xe.numpy())

    # ---- DroidLoss (losses.py:304-342): y = x + t z, u = z mask ----
    mask = make_mask(B, N, "f32", gen)
    x = torch.randn(B, N, Fe, generator=gen) * mask
    cond = torch.randn(B, Cg, generator=gen)
    loss_mod = ref.losses.DroidLoss(flows=flows, sigma=1e-4)
    torch.manual_seed(777)
    cnf.zero_grad()
    loss = loss_mod(x, mask=mask, cond=cond)
    loss.backward()
    torch.manual_seed(777)  # losses.py:330, 335
    t = torch.rand_like(torch.ones(B))
    z = torch.randn_like(x)
    tag = "droid/"
    out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
    out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
    for k, p in [kp for kp in cnf.named_parameters() if kp[1].grad is not None][:6]:
        out[tag + "grad/flows.0." + k] = subsample(p.grad.detach().clone().numpy())
    path = os.path.join(out_dir, f"epicw_{name}.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# model "droid_fullcrossattention" (configs/model/fm_droid_crossattention.yaml): same recorder as the transformer's
# ----------------------------------------------------------------------------------------------
def ca_net_config(model_dim, num_layers, num_heads, hddn):
    return dict(
        node_embd_config=dict(act_h="lrlu", nrm="layer"),
        ctxt_embd_config=dict(outp_dim=64, act_h="lrlu", nrm="layer"),
        cae_config=dict(model_dim=model_dim, num_layers=num_layers,
                        mha_config=dict(num_heads=num_heads, init_zeros=True, do_layer_norm=True),
                        dense_config=dict(hddn_dim=hddn, act_h="lrlu", nrm="layer", output_init_zeros=True)),
        outp_embd_config=dict(act_h="lrlu", nrm="layer", output_init_zeros=True),
    )


CA_BASE = dict(model="droid_fullcrossattention", features=3, frequencies=16, add_time_to_input=True, t_emb="cosine",
               loss_type="FM-OT")
CA_CONFIGS = {
    "small": (dict(CA_BASE, num_particles=40, global_cond_dim=3, net_config=ca_net_config(128, 2, 16, 256)), 4, False),
    # the yaml's own sizes with experiment/lhco/jets_crossattention.yaml:28-29
    "lhco": (dict(CA_BASE, num_particles=279, global_cond_dim=5, net_config=ca_net_config(128, 8, 16, 256)), 2, False),
    "gauss": (dict(CA_BASE, num_particles=24, global_cond_dim=2, t_emb="gaussian", hidden_dim=64, net_config=ca_net_config(128, 1, 16, 256)), 3, False),
    "plain": (dict(CA_BASE, num_particles=20, global_cond_dim=0, t_emb="sincos", frequencies=6, net_config=ca_net_config(128, 1, 16, 256)), 3, False),
}


# ----------------------------------------------------------------------------------------------
# model="mdma" (configs/model/flow_matching_mdma.yaml:15-36; experiment/jetnet/fm_mdma.yaml, calo_challenge/fm_mdma.yaml).
# CNF passes input_dim and **net_config to MDMA (flow_matching_module.py:163-167): its own global_cond_dim is net_config's (0).
# ----------------------------------------------------------------------------------------------
def mdma_net_config(layers, hidden=128, latent=16):
    return dict(feats=3, latent=latent, layers=layers, hidden_dim=hidden, activation="leaky_relu", wrapper_func="weight_norm",
                frequencies=6, num_points=150, t_local_cat=False, t_global_cat=False, global_cond_dim=0, local_cond_dim=0,
                dropout=0.0, sum_scale=1e-2)


MDMA_BASE = dict(model="mdma", features=3, frequencies=16, add_time_to_input=True, t_emb="cosine", loss_type="FM-OT",
                 global_cond_dim=0)
MDMA_CONFIGS = {
    "small": (dict(MDMA_BASE, num_particles=40, net_config=mdma_net_config(2)), 4, False),
    # the yaml's own sizes with experiment/jetnet/fm_mdma.yaml:27 (150 particles) and calo_challenge/fm_mdma.yaml:26 (4 features)
    "yaml": (dict(MDMA_BASE, num_particles=150, features=4, net_config=mdma_net_config(4)), 2, False),
    # MDMA.__init__'s own defaults (mdma.py:101-102): the time embedding concatenated to the particle and to the class-token Linears
    # (net_config.frequencies must then be the CNF's: the Linears are sized by it, the embedding by the CNF's)
    "tcat": (dict(MDMA_BASE, num_particles=40, net_config=dict(mdma_net_config(2), frequencies=16, t_local_cat=True, t_global_cat=True)), 3, False),
    "tloc": (dict(MDMA_BASE, num_particles=24, add_time_to_input=False, frequencies=6,
                  net_config=dict(mdma_net_config(1), frequencies=6, t_local_cat=True, t_global_cat=False)), 3, False),
    # t_emb="gaussian" (flow_matching_module.py:178-181, 213-221) in front of an MDMA that concatenates the embedding everywhere
    "gauss": (dict(MDMA_BASE, num_particles=24, t_emb="gaussian", hidden_dim=64, frequencies=6,
                   net_config=dict(mdma_net_config(2), frequencies=6, t_local_cat=True, t_global_cat=True)), 3, False),
    # the conditional variant (mdma.py:60-63, 79-82, 157-174: one condition value per jet, global_cond_in (B, 1)): as the class token's and
    # `cond`'s extra input (net_config.global_cond_dim = 1), appended to the class-token Linears (global_cat_cond) and to the particle
    # Linears (local_cat_cond; without global_cond_dim the blocks append the particle COUNT, cond[..., -1:], the ends the condition)
    "cond": (dict(MDMA_BASE, num_particles=24, global_cond_dim=1, net_config=dict(mdma_net_config(2), global_cond_dim=1)), 3, False),
    "condcat": (dict(MDMA_BASE, num_particles=24, global_cond_dim=1, frequencies=6,
                     net_config=dict(mdma_net_config(2), frequencies=6, global_cond_dim=1, t_local_cat=True, t_global_cat=True,
                                     local_cat_cond=True, global_cat_cond=True)), 3, False),
    "lcat": (dict(MDMA_BASE, num_particles=24, global_cond_dim=1, add_time_to_input=False, frequencies=6,
                  net_config=dict(mdma_net_config(1), frequencies=6, local_cat_cond=True)), 3, False),
    "tglob": (dict(MDMA_BASE, num_particles=24, add_time_to_input=False, frequencies=6, t_emb="sincos",
                   net_config=dict(mdma_net_config(1), frequencies=6, t_local_cat=False, t_global_cat=True)), 3, False),
}


# ----------------------------------------------------------------------------------------------
# loss_type="diffusion" (configs/model/diffusion.yaml: EPiC, hidden 128, cosine embedding, huber criterion)
# ----------------------------------------------------------------------------------------------
DIFF_HP = dict(BASE, num_particles=30, layers=2, global_cond_dim=2, loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02})


def gen_diffusion(ref, out_dir, B=4, seed=2468):
    import json

    hp = DIFF_HP
    torch.manual_seed(seed)
    cnf = ref.fmm.CNF(**hp)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in cnf.named_parameters():
            if k.endswith("weight_g"):
                p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif k.endswith("bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=gen))
    flows = torch.nn.ModuleList([cnf])
    state = {f"flows.0.{k}": v.detach().clone() for k, v in cnf.state_dict().items()}
    N, Fe, Cg, dc = hp["num_particles"], hp["features"], hp["global_cond_dim"], hp["diff_config"]
    out = {"_keys": np.array(list(state.keys()))}
    for k, v in state.items():
        out["sd/" + k] = v.numpy()
    out["hp_json"] = np.array(json.dumps(hp))
    out["freqs"] = torch.arange(2 * hp["frequencies"]).exp().numpy()
    # ---- DiffusionLoss (losses.py:207-290), both criteria; draws replayed by seed (:241, :247) ----
    for crit in ("huber", "mse"):
        mask = make_mask(B, N, "f32", gen)
        x = 2.0 * torch.randn(B, N, Fe, generator=gen) * mask  # scale 2: some residuals beyond the huber knee
        cond = torch.randn(B, Cg, generator=gen)
        loss_mod = ref.losses.DiffusionLoss(flows=flows, criterion=crit, diff_config=dc)
        torch.manual_seed(1357)
        cnf.zero_grad()
        loss = loss_mod(x, mask=mask, cond=cond)
        loss.backward()
        torch.manual_seed(1357)
        t = torch.rand_like(torch.ones(B))
        z = torch.randn_like(x) * mask
        tag = f"loss_{crit}/"
        out[tag + "x"], out[tag + "t"], out[tag + "z"] = x.numpy(), t.numpy(), z.numpy()
        out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
        for k, p in cnf.named_parameters():
            out[tag + "grad/flows.0." + k] = p.grad.detach().clone().numpy()
    # ---- probability-flow ODE right-hand side (ode_wrapper, flow_matching_module.py:62-69) and the midpoint sampler on it ----
    mask = make_mask(B, N, "f32", gen)
    z = torch.randn(B, N, Fe, generator=gen)
    cond = torch.randn(B, Cg, generator=gen)
    wrapped = ref.fmm.ode_wrapper(model=cnf, mask=mask, cond=cond, loss_type="diffusion", diff_config=dc)
    with torch.no_grad():
        t0 = torch.tensor(0.37)
        out["rhs/t"], out["rhs/x"], out["rhs/mask"], out["rhs/cond"] = t0.numpy(), (z * mask).numpy(), mask.numpy(), cond.numpy()
        out["rhs/f"] = wrapped(t0, z * mask).numpy()
        for steps in (3, 10):
            xe = midpoint_trajectory_end(wrapped, z * mask, torch.linspace(1.0, 0.0, steps))
            tag = f"midpoint_{steps}/"
            out[tag + "z"], out[tag + "mask"], out[tag + "cond"], out[tag + "x_end"] = z.numpy(), mask.numpy(), cond.numpy(), xe.numpy()
    # ---- DDIM and Euler-Maruyama samplers (solver.py:22-143) ----
    sched = ref.diffusion.VPDiffusionSchedule(**dc)
    n_steps = 6
    x_ddim, _ = ref.solver.ddim_sampler(cnf, sched, (z * mask).clone(), n_steps=n_steps, mask=mask, cond=cond)
    out["ddim/z"], out["ddim/mask"], out["ddim/cond"], out["ddim/x_end"] = z.numpy(), mask.numpy(), cond.numpy(), x_ddim.numpy()
    torch.manual_seed(8642)
    x_em, _ = ref.solver.euler_maruyama_sampler(cnf, sched, (z * mask).clone(), n_steps=n_steps, mask=mask, cond=cond)
    torch.manual_seed(8642)
    noise = torch.stack([torch.randn_like(z) for _ in range(n_steps)])  # solver.py:131, the sampler's only draws
    out["em/z"], out["em/mask"], out["em/cond"], out["em/noise"], out["em/x_end"] = (
        z.numpy(), mask.numpy(), cond.numpy(), noise.numpy(), x_em.numpy())
    out["n_steps"] = np.array(n_steps)
    path = os.path.join(out_dir, "epic_diffusion.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# loss_type="diffusion" on the other model paths (flow_matching_module.py:452-458 builds DiffusionLoss for any `model`): the reduced
# Full-Transformer / cross-attention / MDMA configurations with seed-derived weights (oracle/seeded.py), gradients sub-sampled
# ----------------------------------------------------------------------------------------------
DIFF_ROWS_CONFIGS = {
    "tf": (lambda: dict(TF_CONFIGS["small"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 3131),
    "ca": (lambda: dict(CA_CONFIGS["small"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 4242),
    "mdma": (lambda: dict(MDMA_CONFIGS["small"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 5353),
    # ... with t_emb="gaussian" (the CNF builds both for any model): every model path, both EPiC paths (files <model>_diffusion_gauss.npz)
    "tf_gauss": (lambda: dict(TF_CONFIGS["gauss"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 3132),
    "ca_gauss": (lambda: dict(CA_CONFIGS["gauss"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 4243),
    "mdma_gauss": (lambda: dict(MDMA_CONFIGS["gauss"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 5354),
    "epic_gauss": (lambda: dict(CONFIGS["gauss"], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 6465),
    "epicw_gauss": (lambda: dict(WIDE_CONFIGS["gauss"][0], loss_type="diffusion", diff_config={"max_sr": 0.999, "min_sr": 0.02}), 7576),
}


def gen_diffusion_rows(ref, prefix, out_dir, B=3):
    import copy
    import json

    from oracle.seeded import seeded_state, subsample

    mk_hp, seed = DIFF_ROWS_CONFIGS[prefix]
    hp = mk_hp()
    torch.manual_seed(seed)
    cnf = ref.fmm.CNF(**copy.deepcopy(hp))
    shapes = {k: tuple(v.shape) for k, v in cnf.state_dict().items() if k != "frequencies"}
    new = seeded_state(shapes, seed)
    sd = cnf.state_dict()
    for k, v in new.items():
        sd[k] = torch.from_numpy(v)
    cnf.load_state_dict(sd)
    flows = torch.nn.ModuleList([cnf])
    N, Fe, Cg, dc = hp["num_particles"], hp["features"], hp["global_cond_dim"], hp["diff_config"]
    out = {"_keys": np.array(["flows.0." + k for k in cnf.state_dict().keys()])}
    out["_shapes_json"] = np.array(json.dumps({"flows.0." + k: list(s) for k, s in shapes.items()}))
    out["seed"] = np.array(seed)
    out["hp_json"] = np.array(json.dumps(hp))
    out["freqs"] = (torch.arange(2 * hp["frequencies"]).exp() if hp["t_emb"] == "cosine" else cnf.frequencies.clone()).numpy()
    out["abs_sum"] = np.array(sum(float(np.abs(v).sum(dtype=np.float64)) for v in new.values()))
    gen = torch.Generator().manual_seed(seed + 1)
    mk_cond = lambda: torch.randn(B, Cg, generator=gen) if Cg else None
    put = lambda tag, **kw: out.update({tag + k: (np.zeros(0, np.float32) if v is None else v.numpy()) for k, v in kw.items()})
    # ---- DiffusionLoss (losses.py:207-290), both criteria; draws replayed by seed (:241, :247) ----
    for crit in ("huber", "mse"):
        mask = make_mask(B, N, "f32", gen)
        x = 2.0 * torch.randn(B, N, Fe, generator=gen) * mask
        cond = mk_cond()
        loss_mod = ref.losses.DiffusionLoss(flows=flows, criterion=crit, diff_config=dc)
        torch.manual_seed(1357)
        cnf.zero_grad()
        loss = loss_mod(x, mask=mask, cond=cond)
        loss.backward()
        torch.manual_seed(1357)
        t = torch.rand_like(torch.ones(B))
        z = torch.randn_like(x) * mask
        tag = f"loss_{crit}/"
        put(tag, x=x, t=t, z=z, mask=mask, cond=cond, loss=loss.detach())
        for k, p in cnf.named_parameters():
            if p.grad is None:  # MDMA: Block.cond_cls is constructed but never used (mdma.py:30, 36)
                continue
            out[tag + "grad/flows.0." + k] = subsample(p.grad.detach().clone().numpy())
    # ---- probability-flow ODE right-hand side (ode_wrapper, flow_matching_module.py:62-69), midpoint on it, DDIM, Euler-Maruyama ----
    mask = make_mask(B, N, "f32", gen)
    z = torch.randn(B, N, Fe, generator=gen)
    cond = mk_cond()
    wrapped = ref.fmm.ode_wrapper(model=cnf, mask=mask, cond=cond, loss_type="diffusion", diff_config=dc)
    with torch.no_grad():
        t0 = torch.tensor(0.37)
        put("rhs/", t=t0, x=z * mask, mask=mask, cond=cond, f=wrapped(t0, z * mask))
        for steps in (3, 10):
            xe = midpoint_trajectory_end(wrapped, z * mask, torch.linspace(1.0, 0.0, steps))
            put(f"midpoint_{steps}/", z=z, mask=mask, cond=cond, x_end=xe)
        sched = ref.diffusion.VPDiffusionSchedule(**dc)
        n_steps = 5
        x_ddim, _ = ref.solver.ddim_sampler(cnf, sched, (z * mask).clone(), n_steps=n_steps, mask=mask, cond=cond)
        put("ddim/", z=z, mask=mask, cond=cond, x_end=x_ddim)
        torch.manual_seed(8642)
        x_em, _ = ref.solver.euler_maruyama_sampler(cnf, sched, (z * mask).clone(), n_steps=n_steps, mask=mask, cond=cond)
        torch.manual_seed(8642)
        noise = torch.stack([torch.randn_like(z) for _ in range(n_steps)])
        put("em/", z=z, mask=mask, cond=cond, noise=noise, x_end=x_em)
    out["n_steps"] = np.array(n_steps)
    path = os.path.join(out_dir, f"{prefix[:-6]}_diffusion_gauss.npz" if prefix.endswith("_gauss") else f"{prefix}_diffusion.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# n_transforms = 2 (flow_matching_module.py:421-443): two EPiC flows; the losses feed the first flow's output to the second at the
# same time t (losses.py:66-69), forward(reverse=True) decodes through the flows in reverse order (:485-487)
# ----------------------------------------------------------------------------------------------
CHAIN_HP = dict(BASE, num_particles=30, layers=2, global_cond_dim=2, local_cond_dim=2)


# the same two-flow chain at a width the jet-resident kernel does not take (hidden 136 -> the row-matrix path, padded to 192)
CHAIN_WIDE_HP = dict(CHAIN_HP, hidden_dim=136, layers=1, num_particles=24, latent=12)


def gen_chain(ref, out_dir, B=4, seed=1212, hp=None, file_name="epic_chain2.npz"):
    import json

    hp = hp or CHAIN_HP
    gen = torch.Generator().manual_seed(seed + 1)
    cnfs = []
    for i in range(2):
        torch.manual_seed(seed + 10 * i)
        cnf = ref.fmm.CNF(**hp)
        with torch.no_grad():
            for k, p in cnf.named_parameters():
                if k.endswith("weight_g"):
                    p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
                elif k.endswith("bias"):
                    p.add_(0.05 * torch.randn(p.shape, generator=gen))
        cnfs.append(cnf)
    flows = torch.nn.ModuleList(cnfs)
    state = {f"flows.{i}.{k}": v.detach().clone() for i, c in enumerate(cnfs) for k, v in c.state_dict().items()}
    N, Fe, Cg = hp["num_particles"], hp["features"], hp["global_cond_dim"]
    out = {"_keys": np.array(list(state.keys()))}
    for k, v in state.items():
        out["sd/" + k] = v.numpy()
    out["hp_json"] = np.array(json.dumps(dict(hp, n_transforms=2)))
    out["freqs"] = torch.arange(2 * hp["frequencies"]).exp().numpy()
    losses = {"fm": (ref.losses.FlowMatchingLoss, 3), "cfm": (ref.losses.ConditionalFlowMatchingLoss, 3)}
    for name, (cls, _) in losses.items():
        mask = make_mask(B, N, "f32", gen)
        x = torch.randn(B, N, Fe, generator=gen) * mask
        cond = torch.randn(B, Cg, generator=gen)
        loss_mod = cls(flows=flows, sigma=1e-4)
        torch.manual_seed(2468)
        flows.zero_grad()
        loss = loss_mod(x, mask=mask, cond=cond)
        loss.backward()
        torch.manual_seed(2468)
        t = torch.rand_like(torch.ones(B))
        a = torch.randn_like(x)
        tag = f"loss_{name}/"
        out[tag + "x"], out[tag + "t"], out[tag + "a"] = x.numpy(), t.numpy(), a.numpy()
        if name == "cfm":
            out[tag + "eps"] = torch.randn_like(x).numpy()  # losses.py:116, the third draw
        out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
        for i, c in enumerate(cnfs):
            for k, p in c.named_parameters():
                out[tag + f"grad/flows.{i}." + k] = p.grad.detach().clone().numpy()
    # ---- the composed field and sampling: forward(reverse=True) = for f in reversed(flows): x = f.decode(x, ...) ----
    mask = make_mask(B, N, "f32", gen)
    z = torch.randn(B, N, Fe, generator=gen)
    cond = torch.randn(B, Cg, generator=gen)
    with torch.no_grad():
        x = z * mask
        for steps in (3, 10):
            xe = x
            for c in reversed(cnfs):
                xe = midpoint_trajectory_end(lambda tt, xx: c(tt, xx, mask=mask, cond=cond), xe, torch.linspace(1.0, 0.0, steps))
            tag = f"midpoint_{steps}/"
            out[tag + "z"], out[tag + "mask"], out[tag + "cond"], out[tag + "x_end"] = z.numpy(), mask.numpy(), cond.numpy(), xe.numpy()
    path = os.path.join(out_dir, file_name)
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# n_transforms = 2 on the Full-Transformer / cross-attention models (flow_matching_module.py:421-443): two flows with seed-derived
# weights, the losses of losses.py:66-69 / 125-128 (each flow's output is the next one's input, same t), sampling through both in reverse
# ----------------------------------------------------------------------------------------------
def gen_chain_rows(ref, prefix, out_dir, B=3, seed=8181):
    import copy
    import json

    from oracle.seeded import seeded_state, subsample

    # "epic_gauss" / "epicw_gauss": both EPiC paths with t_emb="gaussian" (the fields then return d / d temb beside d / d x)
    hp = copy.deepcopy({"tf": lambda: TF_CONFIGS["small"][0], "ca": lambda: CA_CONFIGS["small"][0], "epic_gauss": lambda: CONFIGS["gauss"],
                        "epicw_gauss": lambda: WIDE_CONFIGS["gauss"][0], "epic_diff": lambda: DIFF_HP}[prefix]())
    cnfs = []
    for i in range(2):
        torch.manual_seed(seed + i)
        cnfs.append(ref.fmm.CNF(**copy.deepcopy(hp)))
    shapes = {f"flows.{i}.{k}": tuple(v.shape) for i, c in enumerate(cnfs) for k, v in c.state_dict().items() if k != "frequencies"}
    new = seeded_state(shapes, seed)
    for i, c in enumerate(cnfs):
        sd = c.state_dict()
        for k in list(sd):
            if k != "frequencies":
                sd[k] = torch.from_numpy(new[f"flows.{i}.{k}"])
        c.load_state_dict(sd)
    flows = torch.nn.ModuleList(cnfs)
    N, Fe, Cg = hp["num_particles"], hp["features"], hp["global_cond_dim"]
    out = {"_keys": np.array([f"flows.{i}.{k}" for i, c in enumerate(cnfs) for k in c.state_dict().keys()])}
    out["_shapes_json"] = np.array(json.dumps({k: list(v) for k, v in shapes.items()}))
    out["seed"] = np.array(seed)
    out["hp_json"] = np.array(json.dumps(dict(hp, n_transforms=2)))
    out["freqs"] = (torch.arange(2 * hp["frequencies"]).exp() if hp["t_emb"] == "cosine" else cnfs[0].frequencies.clone()).numpy()
    out["abs_sum"] = np.array(sum(float(np.abs(v).sum(dtype=np.float64)) for v in new.values()))
    gen = torch.Generator().manual_seed(seed + 1)
    diffusion = hp.get("loss_type") == "diffusion"  # "epic_diff": DiffusionLoss through both flows (losses.py:264-267), both criteria
    kinds = (("huber", None), ("mse", None)) if diffusion else (("fm", ref.losses.FlowMatchingLoss), ("cfm", ref.losses.ConditionalFlowMatchingLoss))
    for name, cls in kinds:
        mask = make_mask(B, N, "f32", gen)
        x = torch.randn(B, N, Fe, generator=gen) * mask
        cond = torch.randn(B, Cg, generator=gen)
        if diffusion:
            loss_mod = ref.losses.DiffusionLoss(flows=flows, criterion=name, diff_config=hp["diff_config"])
            torch.manual_seed(2468)
            flows.zero_grad()
            loss = loss_mod(2.0 * x, mask=mask, cond=cond)
            loss.backward()
            torch.manual_seed(2468)
            t = torch.rand_like(torch.ones(B))
            a = torch.randn_like(x) * mask  # losses.py:247
            tag = f"loss_{name}/"
            out[tag + "x"], out[tag + "t"], out[tag + "a"] = (2.0 * x).numpy(), t.numpy(), a.numpy()
            out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
            for i, c in enumerate(cnfs):
                for k, p in c.named_parameters():
                    if p.grad is not None:
                        out[tag + f"grad/flows.{i}." + k] = subsample(p.grad.detach().clone().numpy())
            continue
        loss_mod = cls(flows=flows, sigma=1e-4)
        torch.manual_seed(2468)
        flows.zero_grad()
        loss = loss_mod(x, mask=mask, cond=cond)
        loss.backward()
        torch.manual_seed(2468)
        t = torch.rand_like(torch.ones(B))
        a = torch.randn_like(x)
        tag = f"loss_{name}/"
        out[tag + "x"], out[tag + "t"], out[tag + "a"] = x.numpy(), t.numpy(), a.numpy()
        if name == "cfm":
            out[tag + "eps"] = torch.randn_like(x).numpy()
        out[tag + "mask"], out[tag + "cond"], out[tag + "loss"] = mask.numpy(), cond.numpy(), loss.detach().numpy()
        for i, c in enumerate(cnfs):
            for k, p in c.named_parameters():
                if p.grad is not None:
                    out[tag + f"grad/flows.{i}." + k] = subsample(p.grad.detach().clone().numpy())
    mask = make_mask(B, N, "f32", gen)
    z = torch.randn(B, N, Fe, generator=gen)
    cond = torch.randn(B, Cg, generator=gen)
    with torch.no_grad():
        for steps in (3, 10):
            xe = z * mask
            for c in reversed(cnfs):
                wrapped = ref.fmm.ode_wrapper(model=c, cond=cond, mask=mask, loss_type="diffusion" if diffusion else "FM-OT",
                                              **({"diff_config": hp["diff_config"]} if diffusion else {}))
                xe = midpoint_trajectory_end(wrapped, xe, torch.linspace(1.0, 0.0, steps))
            tag = f"midpoint_{steps}/"
            out[tag + "z"], out[tag + "mask"], out[tag + "cond"], out[tag + "x_end"] = z.numpy(), mask.numpy(), cond.numpy(), xe.numpy()
    fname = f"{prefix[:-6]}_chain2_gauss.npz" if prefix.endswith("_gauss") else ("epic_chain2_diffusion.npz" if diffusion else f"{prefix}_chain2.npz")
    path = os.path.join(out_dir, fname)
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(out)} arrays")


# ----------------------------------------------------------------------------------------------
# IterativeNormLayer (norm_layer.py): three training batches, then eval-mode forward / reverse
# ----------------------------------------------------------------------------------------------
def gen_norm_layer(ref, out_dir, seed=97531):
    gen = torch.Generator().manual_seed(seed)
    B, N, Fe = 8, 12, 3
    layer = ref.norm_layer.IterativeNormLayer((Fe,), max_n=150)
    layer.train()
    out = {"max_n": np.array(150)}
    for k in range(4):
        mask = make_mask(B, N, "f32", gen).squeeze(-1) == 1
        x = (torch.randn(B, N, Fe, generator=gen) * torch.tensor([1.0, 2.5, 0.3]) + torch.tensor([0.5, -1.0, 3.0])) * mask.unsqueeze(-1)
        y = layer(x, mask)
        tag = f"step{k}/"
        out[tag + "x"], out[tag + "mask"], out[tag + "y"] = x.numpy(), mask.numpy(), y.numpy()
        for b in ("means", "vars", "n", "m2"):
            out[tag + b] = getattr(layer, b).detach().clone().numpy()
        out[tag + "frozen"] = np.array(bool(layer.frozen))
    layer.eval()
    out["rev/y"] = layer.reverse(y, mask).numpy()  # of the last batch
    c = torch.randn(B, 2, generator=gen) * 3 + 1
    cl = ref.norm_layer.IterativeNormLayer((2,), max_n=250)
    cl.train()
    out["cond/x"], out["cond/y"] = c.numpy(), cl(c).numpy()
    for b in ("means", "vars", "n", "m2"):
        out["cond/" + b] = getattr(cl, b).detach().clone().numpy()
    path = os.path.join(out_dir, "norm_layer.npz")
    np.savez(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)/1e3:.1f} kB, {len(out)} arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of {epic,no_sets,tf,wide,ca,mdma,diffusion,diffusion_rows,chain,chain_wide,chain_rows,norm}; default all")
    ap.add_argument("--names", default="", help="with --only epic / wide / tf / ca / mdma: comma list of configuration names (default all)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    ap2 = args.only.split(",") if args.only else None
    names = args.names.split(",") if args.names else None
    for name, hp in CONFIGS.items():
        if (ap2 is None or "epic" in ap2) and (names is None or name in names):
            gen_config(ref, name, hp, args.out)
    if ap2 is None or "no_sets" in ap2:
        gen_no_sets(ref, args.out)
    if ap2 is None or "diffusion" in ap2:
        gen_diffusion(ref, args.out)
    if ap2 is None or "norm" in ap2:
        gen_norm_layer(ref, args.out)
    if ap2 is None or "chain" in ap2:
        gen_chain(ref, args.out)
    for prefix in ("tf", "ca", "epic_gauss", "epicw_gauss", "epic_diff"):
        if (ap2 is None or "chain_rows" in ap2) and (names is None or prefix in names):
            gen_chain_rows(ref, prefix, args.out)
    if ap2 is None or "chain_wide" in ap2:
        gen_chain(ref, args.out, B=3, seed=3434, hp=CHAIN_WIDE_HP, file_name="epic_chain2w.npz")
    for prefix in DIFF_ROWS_CONFIGS:
        if (ap2 is None or "diffusion_rows" in ap2) and (names is None or prefix in names):
            gen_diffusion_rows(ref, prefix, args.out)
    for name, (hp, B) in WIDE_CONFIGS.items():
        if (ap2 is None or "wide" in ap2) and (names is None or name in names):
            gen_epic_wide(ref, name, hp, B, args.out)
    for name, (hp, B, store_all) in TF_CONFIGS.items():
        if (ap2 is None or "tf" in ap2) and (names is None or name in names):
            gen_transformer(ref, name, hp, B, store_all, args.out)
    for name, (hp, B, store_all) in CA_CONFIGS.items():
        if (ap2 is None or "ca" in ap2) and (names is None or name in names):
            gen_transformer(ref, name, hp, B, store_all, args.out, seed=4048, file_prefix="ca")
    for name, (hp, B, store_all) in MDMA_CONFIGS.items():
        if (ap2 is None or "mdma" in ap2) and (names is None or name in names):
            gen_transformer(ref, name, hp, B, store_all, args.out, seed=6061, file_prefix="mdma")


if __name__ == "__main__":
    main()
