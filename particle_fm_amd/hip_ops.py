"""Thin tensor-level wrappers over the C ABI (device pointers + current HIP stream).

PyTorch is plumbing here: it owns device memory and the stream; every number comes out of
libpfm_hip.so.  All functions require ROCm device tensors and raise otherwise (no CPU path).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .layout import EpicLayout


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_f32(name: str, t: Optional[torch.Tensor], device, shape=None) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the ROCm device (got {t.device}); the HIP path has no CPU fallback")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _prep_common(layout: EpicLayout, blob, x, cond, mask):
    cfg = layout.cfg
    if not x.is_cuda:
        raise RuntimeError("HIP backend needs tensors on a ROCm device; there is no CPU fallback")
    dev = x.device
    B, N, F = x.shape
    if N != cfg.num_particles or F != cfg.features:
        raise ValueError(f"x has shape {tuple(x.shape)}, model expects (*, {cfg.num_particles}, {cfg.features})")
    blob = _dev_f32("blob", blob, dev, (layout.blob_total,))
    x = _dev_f32("x", x, dev)
    if cfg.global_cond_dim > 0 or cfg.local_cond_dim > 0:
        if cond is None:
            raise ValueError("global_cond_dim/local_cond_dim > 0 but no cond given")  # epic.py:313-317
        cond = _dev_f32("cond", cond, dev, (B, cfg.global_cond_dim))
    else:
        cond = None
    if mask is not None:
        mask = _dev_f32("mask", mask.reshape(B, N), dev, (B, N))
    return dev, B, blob, x, cond, mask


def epic_forward(layout: EpicLayout, blob: torch.Tensor, t: torch.Tensor, x: torch.Tensor,
                 cond: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """v = EPiC(t, x, cond, mask).  t: (B,) one time per jet."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t = _dev_f32("t", t, dev, (B,))
    v = torch.empty_like(x)
    rc = lib.pfm_epic_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), _ptr(x), _ptr(cond), _ptr(mask),
                              _ptr(v), B, _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_forward")
    return v


def epic_forward_temb(layout: EpicLayout, blob: torch.Tensor, temb: torch.Tensor, x: torch.Tensor,
                      cond: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """v = EPiC(temb, x, cond, mask) with the time embedding given: temb (B,T)."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    temb = _dev_f32("temb", temb, dev, (B, layout.cfg.t_dim))
    v = torch.empty_like(x)
    rc = lib.pfm_epic_forward_temb(ctypes.byref(layout.desc), _ptr(blob), _ptr(temb), _ptr(x), _ptr(cond),
                                   _ptr(mask), _ptr(v), B, _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_forward_temb")
    return v


def _midpoint_grid_host(ode_steps: int):
    t_span = torch.linspace(1.0, 0.0, ode_steps)
    t = t_span[0]
    dt = t_span[1] - t
    ts, dts = [], []
    for k in range(1, ode_steps):
        ts += [t, t + 0.5 * dt]
        dts.append(dt)
        t = t + dt
        if k < ode_steps - 1:
            dt = t_span[k + 1] - t
    return torch.stack(ts), torch.stack(dts)


_GRID_CACHE: dict = {}


def midpoint_grid(ode_steps: int, device=None):
    """Times and step sizes the fixed-step driver visits for t_span = linspace(1, 0, ode_steps)
    (flow_matching_module.py:285; torchdyn driver: t += dt; dt = t_span[k+1] - t), in fp32 on the host.  The grid is a
    pure function of ``ode_steps``: it is built once (198 scalar tensor ops) and, with ``device``, copied there once --
    a sampler call then costs no host arithmetic and no pageable H2D copy.  Callers must not write to the result."""
    key = (int(ode_steps), None if device is None else str(device))
    hit = _GRID_CACHE.get(key)
    if hit is None:
        host = _GRID_CACHE.get((int(ode_steps), None))
        if host is None:
            host = _GRID_CACHE[(int(ode_steps), None)] = _midpoint_grid_host(int(ode_steps))
        hit = host if device is None else tuple(a.to(device) for a in host)
        if len(_GRID_CACHE) > 64:
            _GRID_CACHE.clear()
        _GRID_CACHE[key] = hit
    return hit


class RkTableau(ctypes.Structure):
    """ctypes mirror of ``pfm_rk_tableau`` (include/pfm_hip.h)."""

    _fields_ = [("stages", ctypes.c_int32), ("pad_", ctypes.c_int32), ("c", ctypes.c_float * 4),
                ("a", (ctypes.c_float * 4) * 4), ("b", ctypes.c_float * 4)]


# The fixed-step torchdyn solvers CNF.decode / CNF.encode name (flow_matching_module.py:235-243, 261-287), as explicit
# Runge-Kutta tableaus (c, rows of a, b).  torchdyn's "rk4" is the 3/8 rule (its construct_rk4), not the classical scheme.
RK_TABLEAUS = {
    "euler": ([0.0], [[]], [1.0]),
    "midpoint": ([0.0, 0.5], [[], [0.5]], [0.0, 1.0]),
    "rk4": ([0.0, 1 / 3, 2 / 3, 1.0], [[], [1 / 3], [-1 / 3, 1.0], [1.0, -1.0, 1.0]], [1 / 8, 3 / 8, 3 / 8, 1 / 8]),
}


def rk_tableau(solver: str) -> RkTableau:
    if solver not in RK_TABLEAUS:
        raise NotImplementedError(f"Solver {solver} has no HIP path in this build (fixed-step 'euler', 'midpoint', 'rk4' do).")
    c, a, b = RK_TABLEAUS[solver]
    t = RkTableau()
    t.stages = len(b)
    for i, v in enumerate(c):
        t.c[i] = v
    for i, row in enumerate(a):
        for j, v in enumerate(row):
            t.a[i][j] = v
    for i, v in enumerate(b):
        t.b[i] = v
    return t


def rk_grid(ode_steps: int, solver: str, t0: float = 1.0, t1: float = 0.0):
    """Stage times and step sizes the fixed-step driver visits for t_span = linspace(t0, t1, ode_steps) (torchdyn driver:
    t += dt; dt = t_span[k+1] - t; stage s is evaluated at t + c[s] * dt), in fp32 on the host (built once per argument set)."""
    key = ("rk", int(ode_steps), solver, float(t0), float(t1))
    if key not in _GRID_CACHE:
        _GRID_CACHE[key] = _rk_grid_host(ode_steps, solver, t0, t1)
    return _GRID_CACHE[key]


def _rk_grid_host(ode_steps: int, solver: str, t0: float = 1.0, t1: float = 0.0):
    c = torch.tensor(RK_TABLEAUS[solver][0], dtype=torch.float32)
    t_span = torch.linspace(t0, t1, ode_steps)
    t = t_span[0]
    dt = t_span[1] - t
    ts, dts = [], []
    for k in range(1, ode_steps):
        ts += [t if s == 0 else t + c[s] * dt for s in range(len(c))]
        dts.append(dt)
        t = t + dt
        if k < ode_steps - 1:
            dt = t_span[k + 1] - t
    return torch.stack(ts), torch.stack(dts)


def epic_sample_rk(layout: EpicLayout, blob: torch.Tensor, z: torch.Tensor, cond: Optional[torch.Tensor] = None,
                   mask: Optional[torch.Tensor] = None, ode_steps: int = 100, solver: str = "rk4", t0: float = 1.0,
                   t1: float = 0.0, diff_config=None, temb_fn=None) -> torch.Tensor:
    """x(t1) from x(t0) = z*mask with the fixed-step explicit Runge-Kutta scheme ``solver`` over linspace(t0, t1, ode_steps),
    one persistent launch.  ``diff_config`` (loss_type="diffusion"): integrate the probability-flow ODE
    -0.5 beta (x - f / noise_rate) of a noise-predicting network instead (ode_wrapper.forward, flow_matching_module.py:62-69)."""
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep_common(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    tab = rk_tableau(solver)
    ts, dts = rk_grid(ode_steps, solver, t0, t1)
    ts, dts = ts.to(dev), dts.to(dev)
    out = torch.empty_like(z)
    # stage slopes | jet order | time-term table of every stage time (the lean evaluation of csrc/epic_fast.h needs the last part)
    kbuf = torch.empty(lib.pfm_epic_sample_rk_scratch_floats(ctypes.byref(layout.desc), tab.stages, ode_steps - 1, B), device=dev,
                       dtype=torch.float32)
    rhs = None
    if diff_config is not None:
        _, nr, beta = diffusion_schedule(ts, **diff_config)
        rhs = torch.stack([-0.5 * beta, nr], dim=1).contiguous()
    if temb_fn is not None:  # caller-supplied embedding of the stage times (t_emb="gaussian")
        if rhs is not None:
            raise NotImplementedError("a caller-supplied time embedding with the diffusion right-hand side")
        tt = _dev_f32("temb_tab", temb_fn(ts), dev, (ts.numel(), layout.cfg.t_dim))
        rc = lib.pfm_epic_sample_rk_temb(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(tt), _ptr(dts), ode_steps - 1,
                                         _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, _ptr(kbuf), _stream_ptr(dev))
        _lib.check(rc, "pfm_epic_sample_rk_temb")
        return out
    rc = lib.pfm_epic_sample_rk_sized(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                                      _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, _ptr(kbuf), kbuf.numel(), _ptr(rhs), _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_sample_rk_sized")
    return out


# ---- two jets per workgroup for sets SHORTER than what the LDS tile could hold (PFM_F_PACK_JETS) --------------------------------
# The kernels size their LDS tile by desc.n_points, so two 30-particle jets never fit the tile of a 30-particle model although the
# CU's LDS would hold ten of them.  With packing asked for, such a call runs on a descriptor of a LARGER set size -- the smallest
# multiple of 16 whose tile takes two full-length jets -- with the inputs zero-padded to it (mask 0 on the padding: those rows are
# never computed); the weights are the same blob with the other descriptor in its tail.  Same numbers, bit for bit: packing never
# mixes rows or per-jet vectors (tests/test_hip_packed.py).
_X2_MASK = 352 + 208 + 128 + 128 + 128 + 128 + 8  # csrc/pfm_common.h: X2_MASK (per-jet vectors of the second jet, in the tail of bufB)


def _seg2_rows(n: int) -> int:
    return n - (_X2_MASK + (n + 3) // 4 * 4 + 127) // 128


QUAD_TILE_ROWS, QUAD_SLOT_ROWS = 128, 32  # csrc/epic_fast.h: four jets per workgroup, each in a fixed 32-row slot of a 128-row tile
import os as _os
# matrix-operand flags that take the quad kernel: PFM_F_BF16_MFMA (11 spilled dwords); the fp32 instantiation spills 48 dwords per
# lane and is opt-in (PFM_QUAD_FP32=1) until measured to pay
QUAD_PRECISIONS = (0, 2) if _os.environ.get("PFM_QUAD_FP32") == "1" else (2,)


def packed_layout(layout: EpicLayout, n: int) -> Optional[EpicLayout]:
    """The descriptor to run an n-particle batch on when jet packing is asked for (PFM_F_PACK_JETS), or None (packing off, the tile
    already takes two full-length jets, or nothing fits).  Sets of <= 32 particles without conditioning go FOUR to a workgroup on the
    128-row tile (PFM_F_QUAD_JETS, bf16 operands); otherwise the smallest tile that takes two full-length jets."""
    from .layout import PFM_F_BF16_MFMA, PFM_F_F16X3_MFMA, PFM_F_GENERIC_SAMPLER, PFM_F_PACK_JETS, PFM_F_QUAD_JETS, PFM_F_SKIP_MASKED_TAIL
    fl = int(layout.desc.flags)
    if not (fl & PFM_F_PACK_JETS) or not (fl & PFM_F_SKIP_MASKED_TAIL):
        return None
    cfg = layout.cfg
    mode = fl & (PFM_F_BF16_MFMA | PFM_F_F16X3_MFMA)
    if (n <= QUAD_SLOT_ROWS and mode in QUAD_PRECISIONS and cfg.global_cond_dim == 0 and cfg.t_dim == 32 and cfg.features <= 4
            and not (fl & PFM_F_GENERIC_SAMPLER)):
        return layout.padded(QUAD_TILE_ROWS, PFM_F_QUAD_JETS)
    tile = packed_tile_rows(layout, n)
    return layout.padded(tile) if tile else None


def packed_tile_rows(layout: EpicLayout, n: int) -> int:
    """Set size to run an n-particle batch on so that two full-length jets share a workgroup, or 0 if packing is off, the tile
    already takes them, or no tile that fits the LDS would."""
    from .layout import PFM_F_PACK_JETS, PFM_F_SKIP_MASKED_TAIL
    fl = int(layout.desc.flags)
    if not (fl & PFM_F_PACK_JETS) or not (fl & PFM_F_SKIP_MASKED_TAIL):
        return 0
    need = (n + 15) // 16 * 16 + n
    if _seg2_rows(n) >= need:
        return 0
    for cand in range((n + 15) // 16 * 16, 161, 16):
        if _seg2_rows(cand) >= need:
            probe = layout.padded(cand)
            return cand if _lib.load().pfm_epic_lds_bytes(ctypes.byref(probe.desc)) <= 163840 - 1152 else 0
    return 0


def epic_sample_midpoint(layout: EpicLayout, blob: torch.Tensor, z: torch.Tensor,
                         cond: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                         ode_steps: int = 100, premask: bool = True, time_table: bool = True,
                         temb_tab: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x(0) from x(1) = z*mask by ode_steps-1 explicit-midpoint intervals, one persistent launch.
    The kernel multiplies the start state by the mask (SetFlowMatchingLitModule.sample does, :668-671);
    ``premask`` is informational: masking twice is idempotent for a 0/1 mask."""
    lib = _lib.load()
    n_in = z.shape[1]
    big = packed_layout(layout, n_in) if time_table else None
    if big is not None:  # run on the larger tile (see packed_layout): padded inputs, the same weights behind the other descriptor
        n_tile = big.cfg.num_particles
        pad = n_tile - n_in
        m2 = mask if mask is not None else torch.ones(z.shape[0], n_in, device=z.device, dtype=torch.float32)
        m2 = m2.reshape(z.shape[0], n_in).to(torch.float32)
        blob_big = torch.cat([blob.reshape(-1)[: int(layout.desc.blob_floats)], big.desc_tail().to(blob.device)])
        out = epic_sample_midpoint(big, blob_big, torch.nn.functional.pad(z, (0, 0, 0, pad)), cond, torch.nn.functional.pad(m2, (0, pad)),
                                   ode_steps=ode_steps, premask=premask, time_table=True, temb_tab=temb_tab)
        return out[:, :n_in].contiguous()
    dev, B, blob, z, cond, mask = _prep_common(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    ts, dts = midpoint_grid(ode_steps, dev)
    out = torch.empty_like(z)
    # time-term table of the call (every jet is evaluated at the same times): cached per (layout, steps, device)
    cache = layout.__dict__.setdefault("_sample_scratch", {})
    key = (ode_steps, str(dev), torch.cuda.current_stream(dev).cuda_stream, B)  # per stream: launches on two streams may overlap
    if key not in cache:
        for k in [k for k in cache if k[:2] != key[:2]]:
            del cache[k]
        cache[key] = torch.empty(max(1, lib.pfm_epic_sample_scratch_floats(ctypes.byref(layout.desc), ode_steps - 1, B)), device=dev,
                                 dtype=torch.float32)
    if temb_tab is not None:  # (2 (ode_steps - 1), T): the caller's embedding of every evaluation time of midpoint_grid
        temb_tab = _dev_f32("temb_tab", temb_tab, dev, (2 * (ode_steps - 1), layout.cfg.t_dim))
        rc = lib.pfm_epic_sample_midpoint_temb(ctypes.byref(layout.desc), _ptr(blob), _ptr(temb_tab), _ptr(dts), ode_steps - 1,
                                               _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B,
                                               _ptr(cache[key] if time_table else None), _stream_ptr(dev))
        _lib.check(rc, "pfm_epic_sample_midpoint_temb")
        return out
    rc = lib.pfm_epic_sample_midpoint(ctypes.byref(layout.desc), _ptr(blob), _ptr(ts), _ptr(dts), ode_steps - 1,
                                      _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B,
                                      _ptr(cache[key] if time_table else None), _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_sample_midpoint")
    return out


JET_ORDER_MIN_JETS = 384  # longest-first launch order of the training kernels from this batch size on (1.5 jets per CU of a 256-CU part)


_ORDER_MEMO: dict = {}


def jet_order(maskf: Optional[torch.Tensor], B: int, n_points: int):
    """order[rank] = jet in descending multiplicity for the loss forward / backward launches (one workgroup per jet, dispatched in
    order: with several jets per CU the short ones should fill the tail).  None for small batches, no mask, or more than 8192 jets."""
    if maskf is None or B < JET_ORDER_MIN_JETS or B > 8192:
        return None
    # the loss forward and its backward ask for the order of the same mask tensor: computed once (the tensor's version counter
    # tells an in-place change; one entry, replaced by the next mask)
    key = (maskf.data_ptr(), maskf._version, B, int(n_points), _stream_ptr(maskf.device).value)
    if _ORDER_MEMO.get("key") == key:
        return _ORDER_MEMO["order"]
    order = torch.empty(B, device=maskf.device, dtype=torch.int32)
    rc = _lib.load().pfm_epic_jet_order(_ptr(maskf), B, int(n_points), _ptr(order), _stream_ptr(maskf.device))
    _lib.check(rc, "pfm_epic_jet_order")
    _ORDER_MEMO.update(key=key, order=order, mask=maskf)  # (keeps the mask alive: its address cannot be reused while it is the key)
    return order


def epic_fm_loss_forward(layout: EpicLayout, blob: torch.Tensor, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor,
                         cond: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                         sigma: float = 1e-4, kind: str = "FM-OT", eps: Optional[torch.Tensor] = None,
                         temb: Optional[torch.Tensor] = None):
    """Flow-matching loss forward with the draws (t, z[, eps]) given; ``temb`` (B,T): the time embedding supplied by the caller
    (t_emb="gaussian") instead of the in-kernel cosine / sincos one.
    Returns (loss_parts (B,), mask_count (B,), saved (B, floats_per_jet))."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t = _dev_f32("t", t, dev, (B,))
    z = _dev_f32("z", z, dev, tuple(x.shape))
    kinds = {"FM-OT": 0, "CFM": 1, "droid": 2}
    if kind not in kinds:
        raise NotImplementedError(f"loss kind {kind} has no HIP kernel")
    if kind == "CFM":
        if eps is None:
            raise ValueError("CFM needs the second noise draw eps")
        if mask is None:
            raise ValueError("CFM loss needs a mask (losses.py:119)")
        eps = _dev_f32("eps", eps, dev, tuple(x.shape))
    per_jet = lib.pfm_epic_saved_floats_per_jet(ctypes.byref(layout.desc))
    saved = torch.empty(B, per_jet, device=dev, dtype=torch.float32)
    parts = torch.empty(B, device=dev, dtype=torch.float32)
    count = torch.empty(B, device=dev, dtype=torch.float32)
    order = jet_order(mask, B, layout.cfg.num_particles)
    if temb is not None:
        temb = _dev_f32("temb", temb, dev, (B, layout.cfg.t_dim))
        rc = lib.pfm_epic_fm_loss_forward_temb(ctypes.byref(layout.desc), _ptr(blob), kinds[kind], float(sigma), _ptr(t), _ptr(temb),
                                               _ptr(x), _ptr(z), _ptr(eps), _ptr(cond), _ptr(mask), _ptr(saved), _ptr(parts),
                                               _ptr(count), B, _ptr(order), _stream_ptr(dev))
        _lib.check(rc, "pfm_epic_fm_loss_forward_temb")
        return parts, count, saved
    rc = lib.pfm_epic_fm_loss_forward(ctypes.byref(layout.desc), _ptr(blob), kinds[kind], float(sigma), _ptr(t),
                                      _ptr(x), _ptr(z), _ptr(eps), _ptr(cond), _ptr(mask), _ptr(saved), _ptr(parts),
                                      _ptr(count), B, _ptr(order), _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_fm_loss_forward")
    return parts, count, saved


BWD_CHUNK_JETS = 2048   # jets per pfm_epic_*_loss_backward call: its scratch is B * (2 layers + 1) * N * 128 floats (1 GiB at 1024 jets
                        # of 150 particles, 6 layers) and the C ABI takes at most 8192 jets per call (DW_MAXB, csrc/epic_dw.h)
_BWD_SCRATCH_SLOTS = 4  # cached scratch buffers per layout (least recently used goes first)


def epic_backward_scratch(layout: EpicLayout, B: int, device) -> torch.Tensor:
    """Scratch of one pfm_epic_*_loss_backward call (gradient rows, per-jet rank-1 operands, partial dW tiles), cached on the
    layout per (B, device, stream): the backward fully writes what it reads, and two backwards on one stream run in order.
    At most _BWD_SCRATCH_SLOTS buffers are kept per layout (LRU), one batch size per (device, stream)."""
    lib = _lib.load()
    cache = layout.__dict__.setdefault("_bwd_scratch", {})
    key = (int(B), str(device), torch.cuda.current_stream(device).cuda_stream)
    if key in cache:
        cache[key] = cache.pop(key)  # most recently used last
        return cache[key]
    n = lib.pfm_epic_backward_scratch_floats(ctypes.byref(layout.desc), int(B))
    if n < 0:
        _lib.check(1, "pfm_epic_backward_scratch_floats")
    for k in [k for k in cache if k[1:] == key[1:]]:  # one batch size per (device, stream) at a time
        del cache[k]
    while len(cache) >= _BWD_SCRATCH_SLOTS:  # streams that are gone (or idle) must not pin their GiB forever
        del cache[next(iter(cache))]
    cache[key] = torch.empty(max(1, n), device=device, dtype=torch.float32)
    return cache[key]


BWD_PHASE_CHAIN, BWD_PHASE_DW = 1, 2  # PFM_BWD_PHASE_* (pfm_hip.h)


def epic_loss_backward_phase(layout: EpicLayout, blob, cond, maskf, saved, inv_total, gscale, gblob, phases: int, *,
                             criterion: Optional[str] = None, jet_w=None) -> None:
    """One half (or both) of the backward, pfm_epic_fm_loss_backward_phases: BWD_PHASE_CHAIN leaves every gradient slot final except
    the 128x128 particle blocks, BWD_PHASE_DW adds those.  One call's worth of jets (B <= BWD_CHUNK_JETS): the data-parallel trainer
    uses it to start the all-reduce of the finished half while the dW GEMM runs."""
    lib = _lib.load()
    dev = blob.device
    B = saved.shape[0]
    if B > BWD_CHUNK_JETS:
        raise ValueError(f"epic_loss_backward_phase: at most {BWD_CHUNK_JETS} jets per call (epic_loss_backward chunks larger batches)")
    scr = epic_backward_scratch(layout, B, dev)
    order = jet_order(maskf, B, layout.cfg.num_particles)
    rc = lib.pfm_epic_fm_loss_backward_phases(ctypes.byref(layout.desc), _ptr(blob), _ptr(cond), _ptr(maskf), _ptr(saved), _ptr(inv_total),
                                              _ptr(gscale), _ptr(gblob), {None: 0, "mse": 0, "huber": 1}[criterion],
                                              _ptr(None if jet_w is None else jet_w), B, _ptr(scr), _ptr(order), int(phases), _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_fm_loss_backward_phases")


def epic_loss_backward(layout: EpicLayout, blob, cond, maskf, saved, inv_total, gscale, gblob, *, criterion: Optional[str] = None,
                       jet_w=None, d_temb=None, d_y=None) -> None:
    """The atomics-free backward of the jet-resident EPiC loss (pfm_epic_fm_loss_backward / _temb / pfm_epic_diffusion_loss_backward):
    WRITES every gradient slot of ``gblob`` (layout.src_gpos) -- nothing has to be zeroed by the caller.  Batches beyond
    BWD_CHUNK_JETS run in chunks (each chunk's gradient is written to a second blob and added in chunk order: still a pure function
    of the inputs), so neither the 8192-jet limit of one call nor its B-proportional scratch bounds the batch size.
    cond / maskf: float32 (B,C) / (B,N) or None; inv_total, gscale: 1-element device tensors; criterion: None (FM-OT / CFM / droid)
    or "mse" / "huber" (diffusion, with jet_w (B,)); d_temb: (B,T) out, or None; d_y: (B,N,F) out (gradient w.r.t. the network's
    particle input, pfm_epic_fm_loss_backward_dx), or None."""
    lib = _lib.load()
    dev = blob.device
    B = saved.shape[0]
    S = _stream_ptr(dev)
    P = _ptr
    tmp = None
    for c0 in range(0, B, BWD_CHUNK_JETS):
        c1 = min(B, c0 + BWD_CHUNK_JETS)
        n = c1 - c0
        out = gblob if c0 == 0 else (tmp if tmp is not None else torch.empty_like(gblob))
        if c0 > 0:
            tmp = out
        cc = None if cond is None else cond[c0:c1]
        mm = None if maskf is None else maskf[c0:c1]
        sv = saved[c0:c1]
        scr = epic_backward_scratch(layout, n, dev)
        order = jet_order(mm, n, layout.cfg.num_particles)
        if criterion is not None:
            rc = lib.pfm_epic_diffusion_loss_backward(ctypes.byref(layout.desc), P(blob), {"mse": 0, "huber": 1}[criterion],
                                                      P(jet_w[c0:c1]), P(cc), P(mm), P(sv), P(inv_total), P(gscale), P(out), n, P(scr),
                                                      P(order), S)
            _lib.check(rc, "pfm_epic_diffusion_loss_backward")
        elif d_y is not None and d_temb is not None:
            rc = lib.pfm_epic_fm_loss_backward_dx_temb(ctypes.byref(layout.desc), P(blob), P(cc), P(mm), P(sv), P(inv_total), P(gscale),
                                                       P(out), P(d_y[c0:c1]), P(d_temb[c0:c1]), n, P(scr), P(order), S)
            _lib.check(rc, "pfm_epic_fm_loss_backward_dx_temb")
        elif d_y is not None:
            rc = lib.pfm_epic_fm_loss_backward_dx(ctypes.byref(layout.desc), P(blob), P(cc), P(mm), P(sv), P(inv_total), P(gscale),
                                                  P(out), P(d_y[c0:c1]), n, P(scr), P(order), S)
            _lib.check(rc, "pfm_epic_fm_loss_backward_dx")
        elif d_temb is not None:
            rc = lib.pfm_epic_fm_loss_backward_temb(ctypes.byref(layout.desc), P(blob), P(cc), P(mm), P(sv), P(inv_total), P(gscale),
                                                    P(out), P(d_temb[c0:c1]), n, P(scr), P(order), S)
            _lib.check(rc, "pfm_epic_fm_loss_backward_temb")
        else:
            rc = lib.pfm_epic_fm_loss_backward(ctypes.byref(layout.desc), P(blob), P(None), P(cc), P(mm), P(sv), P(inv_total),
                                               P(gscale), P(out), n, P(scr), P(order), S)
            _lib.check(rc, "pfm_epic_fm_loss_backward")
        if c0 > 0:
            gblob.add_(tmp)


# ---- loss_type="diffusion" (models/components/diffusion.py, losses.py:207-290, solver.py) ------------------------------
def diffusion_schedule(t: torch.Tensor, max_sr: float = 1.0, min_sr: float = 1e-2):
    """(signal_rate, noise_rate, beta) of the cosine VP schedule at the diffusion times ``t`` (diffusion.py:21-62); O(len(t))
    scalars computed where ``t`` lives, in the reference's fp32 op order."""
    import math
    start, end = math.acos(max_sr), math.acos(min_sr)
    ang = start + t * (end - start)
    return torch.cos(ang), torch.sin(ang), 2 * (end - start) * torch.tan(ang)


def epic_diffusion_loss_forward(layout: EpicLayout, blob, x, t, z, rates, cond=None, mask=None, criterion: str = "huber"):
    """DiffusionLoss forward with the draws given: rates (B,2) = (signal, noise) rate per jet, z already masked.
    Returns (loss_parts (B,) = sum criterion(v - z) per jet, mask_count (B,), saved)."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t = _dev_f32("t", t, dev, (B,))
    z = _dev_f32("z", z, dev, tuple(x.shape))
    rates = _dev_f32("rates", rates, dev, (B, 2))
    crit = {"mse": 0, "huber": 1}[criterion]
    per_jet = lib.pfm_epic_saved_floats_per_jet(ctypes.byref(layout.desc))
    saved = torch.empty(B, per_jet, device=dev, dtype=torch.float32)
    parts = torch.empty(B, device=dev, dtype=torch.float32)
    count = torch.empty(B, device=dev, dtype=torch.float32)
    rc = lib.pfm_epic_diffusion_loss_forward(ctypes.byref(layout.desc), _ptr(blob), crit, _ptr(rates), _ptr(t), _ptr(x), _ptr(z),
                                             _ptr(cond), _ptr(mask), _ptr(saved), _ptr(parts), _ptr(count), B,
                                             _ptr(jet_order(mask, B, layout.cfg.num_particles)), _stream_ptr(dev))
    _lib.check(rc, "pfm_epic_diffusion_loss_forward")
    return parts, count, saved


def diffusion_update_(mode: str, x: torch.Tensor, pred: torch.Tensor, coefs, noise: Optional[torch.Tensor] = None,
                      data_out: Optional[torch.Tensor] = None) -> None:
    """One step of the reference's diffusion samplers, in place on ``x`` (solver.py:81-93, 126-132).
    "ddim": data = (x - c0 pred) / c1; x <- c2 data + c3 pred  (c = noise rate, signal rate, next signal rate, next noise rate)
    "em":   x <- x + 0.5 c1 (x + 2 (-pred / c0)) c2;  x <- x + c3 noise  (c = noise rate, beta, delta_t, sqrt(beta delta_t))"""
    lib = _lib.load()
    dev = x.device
    c = [float(v) for v in coefs]
    rc = lib.pfm_diffusion_update({"ddim": 0, "em": 1}[mode], _ptr(x), _ptr(pred), _ptr(noise), c[0], c[1], c[2], c[3],
                                  _ptr(data_out), ctypes.c_int64(x.numel()), _stream_ptr(dev))
    _lib.check(rc, "pfm_diffusion_update")
