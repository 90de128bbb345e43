"""EPiC network: parameter containers with the reference's state_dict layout + the HIP evaluation.

Mirrors particle_fm/models/components/epic.py:17-391 (EPiC_layer, EPiC_encoder): same constructor
arguments, same parameter names (``fc_*.bias / weight_g / weight_v`` -- old-style
``nn.utils.weight_norm`` on every Linear), same initialisation stream, same ``forward(t_in, x_local,
global_cond_in, mask)`` signature and error behaviour.  The arithmetic runs in libpfm_hip.so; there is
no PyTorch fallback: calling ``forward`` with CPU tensors raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

import ctypes

from ... import _lib, fm_loss, fm_loss_wide, hip_ops, hip_ops_wide
from ...layout import PFM_HIDDEN, EpicConfig, EpicDesc, EpicLayout
from ...freq_table import FreqTableMixin
from ...layout_wide import EpicWideLayout

LDS_BYTES = 160 * 1024   # per CU on gfx950
MAX_RESIDENT_POINTS = 160  # the particle phases of the jet-resident kernel are unrolled for 5 pairs of 16-row tiles


def jet_resident_fits(num_points: int, features: int) -> bool:
    """Does one jet's (N x 128) activation tile pair fit the 160 KiB LDS of a CU, in the inference / loss-forward kernels AND in the
    backward kernel (asked of the library: the carve is defined there, csrc/pfm_common.h, epic_bwd.h)?  N <= 150 at 3 features."""
    if num_points > MAX_RESIDENT_POINTS:
        return False
    d = EpicDesc()
    d.n_points, d.features = int(num_points), int(features)
    lib = _lib.load()
    return max(lib.pfm_epic_lds_bytes(ctypes.byref(d)), lib.pfm_epic_backward_lds_bytes(ctypes.byref(d))) <= LDS_BYTES


class WNLinear(nn.Module):
    """Parameters of ``nn.utils.weight_norm(nn.Linear(in, out))``: bias, weight_g (out,1), weight_v (out,in),
    registered in that order and initialised exactly like the reference (nn.Linear's default init, then
    g = ||v|| row-wise)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        lin = nn.Linear(in_features, out_features)  # consumes the RNG like the reference does
        w = lin.weight.detach()
        self.in_features, self.out_features = in_features, out_features
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(w.clone())

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, weight_norm=True"


def activation_slope(activation: str) -> float:
    """The reference applies ``getattr(F, activation, lambda x: x)`` (epic.py:180): the kernels' activation is max(x, slope * x) with the
    slope in the descriptor -- 0.01 for "leaky_relu" (F.leaky_relu's default, every shipped config), 0 for "relu", 1 (no activation)
    for a name torch.nn.functional does not know, as there; any other function of torch.nn.functional has no HIP path."""
    if activation == "leaky_relu":
        return 0.01
    if activation == "relu":
        return 0.0
    if not hasattr(torch.nn.functional, activation):
        return 1.0
    raise NotImplementedError(f"activation={activation!r}: the HIP kernels implement max(x, slope x): leaky_relu, relu, or none")


def _check_supported(activation: str, wrapper_func: str, dropout: float):
    activation_slope(activation)
    if wrapper_func != "weight_norm":
        raise NotImplementedError(f"wrapper_func={wrapper_func!r}: the HIP path expects weight_norm Linears")
    if dropout != 0.0:
        raise NotImplementedError("dropout > 0 is not implemented in the HIP kernels (all shipped configs use 0.0)")


class EPiC_layer(nn.Module):
    """Parameter container of one EPiC layer (epic.py:37-83); evaluated inside EPiC_encoder's kernel."""

    def __init__(self, local_in_dim: int = 3, hid_dim: int = 256, latent_dim: int = 16, global_cond_dim: int = 0,
                 local_cond_dim: int = 0, t_local_cat: bool = False, t_global_cat: bool = False,
                 activation: str = "leaky_relu", wrapper_func: str = "weight_norm", frequencies: int = 6,
                 num_points: int = 30, dropout: float = 0.0, sum_scale: float = 1e-2):
        super().__init__()
        _check_supported(activation, wrapper_func, dropout)
        tl = 2 * frequencies if t_local_cat else 0
        tg = 2 * frequencies if t_global_cat else 0
        self.fc_global1 = WNLinear(2 * hid_dim + latent_dim + tg + global_cond_dim, hid_dim)
        self.fc_global2 = WNLinear(hid_dim + tg + global_cond_dim, latent_dim)
        self.fc_local1 = WNLinear(local_in_dim + latent_dim + tl + local_cond_dim, hid_dim)
        self.fc_local2 = WNLinear(hid_dim + tl + local_cond_dim, hid_dim)

    def forward(self, *args, **kwargs):
        raise RuntimeError("EPiC_layer is evaluated by the fused EPiC_encoder kernel, not on its own")


class EPiC_encoder(FreqTableMixin, nn.Module):
    """epic.py:206-391.  ``forward(t_in, x_local, global_cond_in, mask)`` with t_in the (B,N,T) time
    embedding, like the reference; ``vector_field(t, x, cond, mask)`` takes the time itself (B,) and lets
    the kernel embed it (what CNF.forward uses)."""

    def __init__(self, latent: int = 16, input_dim: int = 3, hid_d: int = 256, feats: int = 128,
                 equiv_layers: int = 8, global_cond_dim: int = 0, local_cond_dim: int = 0,
                 activation: str = "leaky_relu", wrapper_func: str = "weight_norm", frequencies: int = 6,
                 num_points: int = 30, t_local_cat: bool = False, t_global_cat: bool = False,
                 dropout: float = 0.0, sum_scale: float = 1e-2, t_emb: str = "cosine"):
        super().__init__()
        _check_supported(activation, wrapper_func, dropout)
        self.neg_slope = activation_slope(activation)
        self.t_emb = t_emb  # what CNF embeds the time with: the kernels embed in place of CNF.time_embedding
        # add_time_to_input (flow_matching_module.py:126, 199-200): CNF hands the network cat(time embedding, x), so fc_l1 has
        # 2 * frequencies more input columns.  The kernels fold them into fc_l1's time block (layout.EpicConfig.add_time_to_input).
        if input_dim not in (feats, feats + 2 * frequencies):
            raise NotImplementedError(f"input_dim={input_dim}: the EPiC HIP kernels take the particle features ({feats}) or, with "
                                      f"add_time_to_input, the time embedding in front of them ({feats + 2 * frequencies})")
        self.add_time_to_input = input_dim != feats
        self.latent, self.input_dim, self.hid_d, self.feats = latent, input_dim, hid_d, feats
        self.equiv_layers, self.global_cond_dim, self.local_cond_dim = equiv_layers, global_cond_dim, local_cond_dim
        self.num_points, self.sum_scale = num_points, sum_scale
        self.t_local_cat, self.t_global_cat, self.frequencies = t_local_cat, t_global_cat, frequencies
        tl = 2 * frequencies if t_local_cat else 0
        tg = 2 * frequencies if t_global_cat else 0
        self.fc_l1 = WNLinear(input_dim + tl + local_cond_dim, hid_d)
        self.fc_l2 = WNLinear(hid_d + tl + local_cond_dim, hid_d)
        self.fc_g1 = WNLinear(2 * hid_d + tg + global_cond_dim, hid_d)
        self.fc_g2 = WNLinear(hid_d + tg + global_cond_dim, latent)
        self.nn_list = nn.ModuleList()
        for _ in range(equiv_layers):
            self.nn_list.append(EPiC_layer(hid_d, hid_d, latent, activation=activation, wrapper_func=wrapper_func,
                                           num_points=num_points, t_global_cat=t_global_cat, t_local_cat=t_local_cat,
                                           global_cond_dim=global_cond_dim, local_cond_dim=local_cond_dim,
                                           frequencies=frequencies, dropout=dropout, sum_scale=sum_scale))
        self.fc_l3 = WNLinear(hid_d + tl + local_cond_dim, feats)
        self._layouts: Dict[int, EpicLayout] = {}
        # hidden 128 and a set that fits the LDS tile (N <= 150 at 3 features): one workgroup per jet, activations resident in
        # LDS (pfm_hip.h).  Any other width (JetClass: 300) or a larger set (LHCO x_jet / y_jet: N = 279, whole_event: 560, all on
        # flow_matching.yaml's hidden 128): the multi-kernel GEMM path over all particles (pfm_epicw.h).  Decided per set size:
        # is_wide(n); `wide` is the answer for the module's own num_points.
        self._wide_cache: Dict[int, bool] = {}
        self.skip_masked_tail = True
        # "bf16": the kernels of the jet-resident path (forward, samplers, loss forward, the backward's dX products) run the
        # particle Linears on bf16 MFMA with fp32 accumulate and fp32 activations (PFM_F_BF16_MFMA) -- what
        # trainer.precision="bf16-mixed" means for this model in the reference.  The dW GEMM stays fp32; the row-matrix
        # path has the same switch for its particle Linears (PFM_EW_F_BF16).
        self.mfma_dtype = "fp32"
        # the midpoint sampler may put two short jets into one workgroup (PFM_F_PACK_JETS, include/pfm_hip.h): same results; worth
        # it for large batches of short jets only, hence opt-in
        self.pack_jets = False
        self._fast_pack = None  # set by engine.FusedFMTrainer: one-launch weight-norm pack from the flat buffer

    def is_wide(self, num_points: Optional[int] = None) -> bool:
        n = int(num_points or self.num_points)
        w = self._wide_cache.get(n)
        if w is None:
            w = self._wide_cache[n] = self.hid_d != PFM_HIDDEN or not jet_resident_fits(n, self.feats)
        return w

    @property
    def wide(self) -> bool:
        return self.is_wide(self.num_points)

    # -- layout / weights ------------------------------------------------------------------------
    def config(self, num_points: Optional[int] = None) -> EpicConfig:
        return EpicConfig(num_particles=num_points or self.num_points, features=self.feats, hidden_dim=self.hid_d,
                          latent=self.latent, layers=self.equiv_layers, frequencies=self.frequencies,
                          t_local_cat=self.t_local_cat, t_global_cat=self.t_global_cat,
                          global_cond_dim=self.global_cond_dim, local_cond_dim=self.local_cond_dim,
                          sum_scale=self.sum_scale, t_emb=self.t_emb, add_time_to_input=self.add_time_to_input,
                          neg_slope=self.neg_slope)

    def layout(self, num_points: Optional[int] = None, temb_given: bool = False) -> EpicLayout:
        """``temb_given`` (row-matrix path): the descriptor with PFM_EW_F_TEMB_GIVEN -- same blob, the entry points take the time
        EMBEDDING through `t` (a t_emb="gaussian" configuration always does; the jet-resident path has pfm_epic_*_temb entry points)."""
        n = num_points or self.num_points
        wide = self.is_wide(n)
        if wide:  # row-matrix GEMM path: PFM_EW_F_F16X3 / PFM_EW_F_BF16 / PFM_EW_F_TEMB_GIVEN
            mode = {"fp32": 0, "f16x3": 1, "bf16": 32}[self.mfma_dtype] | (64 if temb_given else 0)
        else:
            mode = {"fp32": 0, "bf16": 2, "f16x3": 4}[self.mfma_dtype] | (16 if self.pack_jets else 0)
        lay = self._layouts.get((n, mode))
        if lay is None:
            if wide:
                lay = EpicWideLayout(self.config(n), flags=mode)
            else:
                lay = EpicLayout(self.config(n), flags=(1 if self.skip_masked_tail else 0) | mode)
            self._layouts[(n, mode)] = lay
        return lay

    def set_jet_packing(self, on: bool = True) -> None:
        """Two short jets per workgroup in the midpoint sampler (jet-resident path, fp32 / bf16 operands); results unchanged."""
        self.pack_jets = bool(on)

    def set_precision(self, precision) -> None:
        """Accepts Lightning's spellings: "bf16", "bf16-mixed", "bf16-true" -> bf16 MFMA operands (inference and training);
        "f16x3" -> split-fp16 operands (three fp16 MFMAs per product block, fp32-grade accuracy, PFM_F_F16X3_MFMA);
        anything else fp32 MFMA."""
        p = str(precision)
        self.mfma_dtype = "bf16" if p.startswith("bf16") else ("f16x3" if p == "f16x3" else "fp32")

    def source_vector(self, layout: Optional[EpicLayout] = None) -> torch.Tensor:
        """effective weights | biases | freqs | 0 from the live parameters (differentiable)."""
        lay = layout or self.layout()
        return lay.source_vector(dict(self.named_parameters()), "", freqs=self.freq_tensor())

    def packed_weights(self, num_points: Optional[int] = None) -> torch.Tensor:
        """The kernel blob for the current parameter values (no autograd).  Cheap (a few small launches);
        rebuilt on every call, so it can never go stale after an optimizer step, load_state_dict or an EMA
        swap (callbacks/ema.py:145-157), and nothing extra ever appears in state_dict()."""
        wide = self.is_wide(num_points)
        if self._fast_pack is not None and not wide:
            blob = self._fast_pack(num_points or self.num_points)
            if blob is not None:
                return blob
        lay = self.layout(num_points)
        with torch.no_grad():
            pack = fm_loss_wide.pack_blob_from_source if wide else fm_loss.pack_blob_from_source
            return pack(lay, self.source_vector(lay))

    # -- evaluation --------------------------------------------------------------------------------
    def _check_inputs(self, t, x_local, global_cond_in):
        if x_local is None:
            raise ValueError("x_local is None")
        if global_cond_in is None and (self.global_cond_dim > 0 or self.local_cond_dim > 0):
            raise ValueError(f"global_cond_dim is {self.global_cond_dim} and local_cond_dim is"
                             f" {self.local_cond_dim} but no global_cond is given")
        if t is None and (self.t_local_cat or self.t_global_cat):
            raise ValueError(f"t_local_cat is {self.t_local_cat} and t_global_cat is {self.t_global_cat} but no"
                             " t is given")

    def forward(self, t_in: torch.Tensor = None, x_local: torch.Tensor = None,
                global_cond_in: torch.Tensor = None, mask: torch.Tensor = None) -> torch.Tensor:
        self._check_inputs(t_in, x_local, global_cond_in)
        wide = self.is_wide(x_local.shape[1])
        lay = self.layout(x_local.shape[1], temb_given=wide)
        B = x_local.shape[0]
        if self.add_time_to_input and x_local.shape[-1] == self.input_dim:
            # the reference's caller passes cat(time embedding, x) (flow_matching_module.py:199-200); the kernels take the particle
            # features and fold the embedding's columns into fc_l1's time block, so the embedding in front of x must be the one in
            # t_in (it is, in CNF.forward); without t_in it is taken from there
            if t_in is None:
                t_in = x_local[:, 0, : 2 * self.frequencies]
            x_local = x_local[..., 2 * self.frequencies:].contiguous()
        if t_in is None:
            temb = torch.zeros(B, lay.cfg.t_dim, device=x_local.device)
        else:
            temb = t_in[:, 0, :] if t_in.dim() == 3 else t_in  # epic.py:342: one embedding per jet
        if wide:  # the row-matrix path takes the embedding rows through its `t` argument
            return hip_ops_wide.ew_forward(lay, self.packed_weights(x_local.shape[1]), temb.expand(B, -1), x_local, global_cond_in, mask)
        return hip_ops.epic_forward_temb(lay, self.packed_weights(x_local.shape[1]), temb, x_local, global_cond_in, mask)

    def vector_field(self, t: torch.Tensor, x_local: torch.Tensor, global_cond_in: torch.Tensor = None,
                     mask: torch.Tensor = None, blob: torch.Tensor = None) -> torch.Tensor:
        """t: (B,) one time per jet; the cosine embedding is evaluated in the kernel."""
        self._check_inputs(t, x_local, global_cond_in)
        lay = self.layout(x_local.shape[1])
        if blob is None:
            blob = self.packed_weights(x_local.shape[1])
        if self.is_wide(x_local.shape[1]):
            return hip_ops_wide.ew_forward(lay, blob, t, x_local, global_cond_in, mask)
        return hip_ops.epic_forward(lay, blob, t, x_local, global_cond_in, mask)
