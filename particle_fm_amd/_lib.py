"""ctypes binding of libpfm_hip.so (C ABI in include/pfm_hip.h, pfm_tf.h, pfm_epicw.h, pfm_ca.h and pfm_mdma.h).

There is no CPU fallback: if the library is missing or a call fails this raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

from .layout import EpicDesc
from .layout_ca import CaDesc
from .layout_mdma import MdmaDesc
from .layout_tf import TfDesc
from .layout_wide import EwDesc

_PKG = os.path.dirname(os.path.abspath(__file__))
IN_TREE_LIB = os.path.join(_PKG, "libpfm_hip.so")
# PFM_LIB_PATH: diagnostics only (tests/diag A/B timing of library variants), and only together with PFM_DIAG=1 -- a variable left
# over from an A/B session must not swap the library under the tests or bench.py.  The product loads the in-tree build.
LIB_PATH = IN_TREE_LIB
if os.environ.get("PFM_LIB_PATH"):
    if os.environ.get("PFM_DIAG") == "1":
        LIB_PATH = os.environ["PFM_LIB_PATH"]
        import sys
        print(f"[particle_fm_amd] PFM_DIAG=1: loading {LIB_PATH} instead of the in-tree library", file=sys.stderr)
    else:
        import warnings
        warnings.warn("PFM_LIB_PATH is set but PFM_DIAG != 1: ignored, the in-tree libpfm_hip.so is used")

_lib = None

# every symbol the headers under include/ declare: name -> (restype, argtypes)
_fp = c_void_p  # device pointers travel as integers (tensor.data_ptr())
SYMBOLS = {
    "pfm_abi_version": (c_int, []),
    "pfm_epic_pack_a16": (c_int, [POINTER(EpicDesc), c_void_p, c_void_p]),
    "pfm_last_error": (c_char_p, []),
    "pfm_epic_lds_bytes": (c_int64, [POINTER(EpicDesc)]),
    "pfm_epic_backward_lds_bytes": (c_int64, [POINTER(EpicDesc)]),
    "pfm_epic_saved_floats_per_jet": (c_int64, [POINTER(EpicDesc)]),
    "pfm_epic_forward": (c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, c_int32, c_void_p]),
    "pfm_epic_forward_temb": (c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, c_int32, c_void_p]),
    "pfm_epic_sample_midpoint": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_sample_scratch_floats": (c_int64, [POINTER(EpicDesc), c_int32, c_int32]),
    "pfm_epic_sample_is_fast": (c_int, [POINTER(EpicDesc)]),
    "pfm_epic_sample_midpoint_temb": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_sample_rk_temb": (c_int, [POINTER(EpicDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_fm_loss_forward_temb": (
        c_int, [POINTER(EpicDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_fm_loss_backward_temb": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_epic_jet_order": (c_int, [_fp, c_int32, c_int32, _fp, c_void_p]),
    "pfm_epic_fm_loss_forward": (
        c_int, [POINTER(EpicDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_backward_scratch_floats": (c_int64, [POINTER(EpicDesc), c_int32]),
    "pfm_loss_finish": (c_int, [_fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_fm_loss_backward": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_epic_fm_loss_backward_dx": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_epic_fm_loss_backward_dx_temb": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_epic_fm_loss_backward_phases": (
        c_int, [POINTER(EpicDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_int32, _fp, _fp, c_int32, c_void_p]),
    "pfm_optim_step": (
        c_int, [_fp, _fp, _fp, _fp, _fp, _fp, c_int64, c_float, c_float, c_float, c_float, c_float, c_float,
                c_float, c_float, c_int32, c_void_p]),
    "pfm_wn_pack": (c_int, [_fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_wn_unpack_grad": (c_int, [_fp, _fp, _fp, c_int32, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_wn_unpack_grad_set": (c_int, [_fp, _fp, _fp, c_int32, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_sample_epilogue": (c_int, [_fp, _fp, _fp, _fp, c_int32, c_int64, c_int32, c_void_p]),
    "pfm_epic_sample_rk": (c_int, [POINTER(EpicDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_epic_sample_rk_scratch_floats": (c_int64, [POINTER(EpicDesc), c_int32, c_int32, c_int32]),
    "pfm_epic_sample_rk_sized": (c_int, [POINTER(EpicDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_int64, _fp, c_void_p]),
    "pfm_epic_diffusion_loss_forward": (
        c_int, [POINTER(EpicDesc), _fp, c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_epic_diffusion_loss_backward": (
        c_int, [POINTER(EpicDesc), _fp, c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_norm_update": (c_int, [_fp, _fp, c_int64, c_int32, _fp, _fp, _fp, _fp, c_int64, c_void_p]),
    "pfm_norm_apply": (c_int, [_fp, _fp, _fp, c_int64, c_int32, _fp, _fp, c_int32, c_void_p]),
    "pfm_diffusion_update": (c_int, [c_int32, _fp, _fp, _fp, c_float, c_float, c_float, c_float, _fp, c_int64, c_void_p]),
    "pfm_tf_sample_rk": (
        c_int, [POINTER(TfDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    "pfm_ew_sample_rk": (
        c_int, [POINTER(EwDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    "pfm_ca_sample_rk": (
        c_int, [POINTER(CaDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    # include/pfm_tf.h
    "pfm_tf_workspace_floats": (c_int64, [POINTER(TfDesc), c_int32, c_int32]),
    "pfm_tf_forward": (c_int, [POINTER(TfDesc), _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_tf_sample_midpoint": (
        c_int, [POINTER(TfDesc), _fp, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    # include/pfm_epicw.h
    "pfm_ew_workspace_floats": (c_int64, [POINTER(EwDesc), c_int32, c_int32]),
    "pfm_ew_forward": (c_int, [POINTER(EwDesc), _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ew_sample_midpoint": (
        c_int, [POINTER(EwDesc), _fp, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    "pfm_ew_fm_loss_forward": (
        c_int, [POINTER(EwDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ew_backward_scratch_floats": (c_int64, [POINTER(EwDesc), c_int32]),
    "pfm_ew_backward_dtemb": (c_int, [POINTER(EwDesc), _fp, c_int32, _fp, c_void_p]),
    "pfm_ew_fm_loss_backward_dx": (c_int, [POINTER(EwDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_ew_diffusion_loss_forward": (
        c_int, [POINTER(EwDesc), _fp, c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ew_diffusion_loss_backward": (
        c_int, [POINTER(EwDesc), _fp, c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_ew_sample_rk_rhs": (
        c_int, [POINTER(EwDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, _fp, c_void_p]),
    "pfm_ew_fm_loss_backward": (c_int, [POINTER(EwDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    # include/pfm_ca.h
    "pfm_ca_workspace_floats": (c_int64, [POINTER(CaDesc), c_int32, c_int32]),
    "pfm_ca_forward": (c_int, [POINTER(CaDesc), _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ca_sample_midpoint": (
        c_int, [POINTER(CaDesc), _fp, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    "pfm_ca_fm_loss_forward": (
        c_int, [POINTER(CaDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ca_backward_scratch_floats": (c_int64, [POINTER(CaDesc), c_int32]),
    "pfm_ca_backward_dtemb": (c_int, [POINTER(CaDesc), _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_ca_fm_loss_backward": (c_int, [POINTER(CaDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_ca_fm_loss_backward_dx": (c_int, [POINTER(CaDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    # include/pfm_mdma.h
    "pfm_mdma_workspace_floats": (c_int64, [POINTER(MdmaDesc), c_int32, c_int32]),
    "pfm_mdma_backward_scratch_floats": (c_int64, [POINTER(MdmaDesc), c_int32]),
    "pfm_mdma_backward_dtemb": (c_int, [POINTER(MdmaDesc), _fp, c_int32, _fp, c_void_p]),
    "pfm_mdma_forward": (c_int, [POINTER(MdmaDesc), _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_mdma_sample_rk": (
        c_int, [POINTER(MdmaDesc), _fp, c_void_p, _fp, _fp, c_int32, _fp, _fp, _fp, _fp, c_int32, c_int32, _fp, _fp, c_void_p]),
    "pfm_mdma_fm_loss_forward": (
        c_int, [POINTER(MdmaDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_mdma_fm_loss_backward": (c_int, [POINTER(MdmaDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_tf_fm_loss_forward": (
        c_int, [POINTER(TfDesc), _fp, c_int32, c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_tf_backward_scratch_floats": (c_int64, [POINTER(TfDesc), c_int32]),
    "pfm_tf_backward_dtemb": (c_int, [POINTER(TfDesc), _fp, _fp, c_int32, _fp, c_void_p]),
    "pfm_tf_fm_loss_backward": (
        c_int, [POINTER(TfDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
    "pfm_tf_fm_loss_backward_dx": (
        c_int, [POINTER(TfDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int32, _fp, _fp, c_void_p]),
}


def load():
    """dlopen the in-tree library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m particle_fm_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    from .layout import PFM_ABI_VERSION
    if lib.pfm_abi_version() != PFM_ABI_VERSION:
        raise RuntimeError(f"libpfm_hip.so ABI version {lib.pfm_abi_version()} != {PFM_ABI_VERSION} (include/pfm_hip.h): stale library, rebuild with "
                           "`python -m particle_fm_amd.build --force`")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pfm_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
